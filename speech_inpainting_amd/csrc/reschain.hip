// reschain.hip -- a whole ResBlock1 (three dilations chained) of the C = 32 vocoder stage as ONE kernel on the raw fp16
// activation stream (gfx950, wave64, v_mfma_f32_16x16x32_f16).
//
// I_ea/hifi_gan/models.py:36-43: for (c1, c2) in zip(convs1, convs2): xt = lrelu(x); xt = c1(xt); xt = lrelu(xt);
// xt = c2(xt); x = xt + x  -- three (c1, c2) pairs with dilations (1, 3, 5) on c1; :112-118 averages the three resblocks
// (kernel sizes 3 / 7 / 11) of a stage.  As one launch per pair (respair.hip) the full-rate stage (C = 32: one row per
// output sample) read and wrote the activation stream nine times per stage and a 256-row tile carried 48-176 MFMAs per
// wave against a fixed tile load, two epilogues and an output pass: 411 TFLOP/s, 40 % of the HBM roof -- bound by
// neither.  Here one persistent 8-wave workgroup per CU keeps a 768-row tile of the residual stream in LDS through all
// three pairs:
//   X   [768][32] raw fp16   the residual stream x_i of the tile (rows 6 (k - 1) from either edge are final after pair 3)
//   A   [32 + 768 + 32][32]  the MFMA operand: lrelu(x_i) for c1, then t = lrelu(c1 + b1) for c2 (written over it), then
//                            lrelu(x_{i+1}); the margins stay zero (taps reach 25 rows past either end)
//   W1, W2  [k][32][32]      ALL taps of the current c1 / c2.  While c1 runs, W2 of the same pair is stored (its loads were
//                            issued a convolution earlier); while c2 runs, W1 of the next pair (of the next tile after
//                            pair 3: the same weights) -- no weight traffic on the critical path, no extra LDS
// Every convolution is computed on all 768 rows; what a row near the tile edge reads from beyond the tile is garbage, the
// region of valid rows shrinks by (k - 1)(d + 1) / 2 per pair and side, and only rows [H, 768 - H), H = 6 (k - 1), are
// stored: 12-31 % more MFMA work than the pair kernels (18 % at k = 11), a third of their HBM traffic, a ninth of their
// tile loads, and no fp32 output image at all -- the residual add happens in the accumulator layout against X in LDS.
//
// What bounds it (timeline build + tools/ubench/mfma_convloop.hip): with N = 32 a 1 KB activation fragment feeds only two
// MFMAs, so a tap costs every wave 8 ds_read_b128 per 12 MFMAs; next to the MFMA stream these reads reach ~210 B/ns per
// CU (alone they stream at 380-490: tools/ubench/lds_read_rate.hip), which is 13 ns per MFMA and SIMD against 8.7-9.7
// for the matrix pipe alone -- the tap loop runs at that rate whatever the schedule (fragment double buffer or not, wave
// priorities, staggered waves: all within 2 %).  The epilogues were bound by 4-way bank conflicts of 8-byte stores in the
// accumulator layout (16 lanes = one column of 16 rows on 64-byte rows); lanes l and l + 16 now trade halves
// (v_permlane16_swap) and store whole 16-byte chunks.
//
// Arithmetic is exactly that of the pair kernels: the same fp16 operands, taps in the same order into the same fp32
// accumulators (started from the bias), t and every x_i rounded to fp16 once, (acc + x) * alpha (+ previous) in fp32 with one rounding --
// the outputs are bit-identical to three respair launches (tests/test_gpu_respair.py).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (make timeline; tools/exp_vocoder_only.py): per-phase wall-clock totals of wave 0 of every
// workgroup, in 100 MHz ticks: 0 tile staging, 1 c1, 2 c1 epilogue, 3 c2, 4 c2 epilogue, 5 output pass, 6 / 7 wave 0's own c2 epilogue / its wait at
// the barrier that ends it (both included in 4), 8 tiles, 9 workgroups.
#ifdef RPW_TIMELINE
__device__ unsigned long long rc_tl[10];
#define RC_TL_DECL unsigned long long tl_t = wall_clock64(); unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tl_tiles = 0; unsigned long long tl_w = 0;
#define RC_TL_W0 tl_w = wall_clock64();
#define RC_TL_W1(ph) tl_acc[ph] += wall_clock64() - tl_w;
#define RC_TL(ph) { const unsigned long long n_ = wall_clock64(); tl_acc[ph] += n_ - tl_t; tl_t = n_; }
#define RC_TL_TILE ++tl_tiles;
#define RC_TL_FLUSH if (threadIdx.x == 0) { for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&rc_tl[q_], tl_acc[q_]); atomicAdd(&rc_tl[8], tl_tiles); atomicAdd(&rc_tl[9], 1ull); }
extern "C" int si_debug_rc_timeline(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rc_tl), sizeof(rc_tl)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[10] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(rc_tl), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define RC_TL_DECL
#define RC_TL_W0
#define RC_TL_W1(ph)
#define RC_TL(ph)
#define RC_TL_TILE
#define RC_TL_FLUSH
#endif

namespace {

constexpr int RC_C = 32, RC_ROWB = 64, RC_NW = 8, RC_NT = 64 * RC_NW;
constexpr int RC_RT = 6;                                              // 16-row MFMA tiles per wave
constexpr int RC_R = RC_NW * RC_RT * 16;                              // 768 rows per tile
constexpr int RC_MARG = 32;                                           // >= 5 * 5 rows a tap reaches past the tile
constexpr int RC_KMAX = 11;
constexpr int RC_XBYTES = RC_R * RC_ROWB;
constexpr int RC_ABYTES = (RC_R + 2 * RC_MARG) * RC_ROWB;
constexpr int RC_WBYTES = RC_KMAX * RC_C * RC_ROWB;                   // one convolution, all taps: 22 KB
constexpr int RC_LDS = RC_XBYTES + RC_ABYTES + 2 * RC_WBYTES + 6 * RC_C * 4;
constexpr int RC_WSLOTS = (RC_WBYTES / 16 + RC_NT - 1) / RC_NT;       // 3
constexpr int RC_YSLOTS = RC_R * 4 / RC_NT;                           // 6
static_assert(RC_R * 4 % RC_NT == 0, "tile staging map");

// 64-byte rows: chunk c of row r lives at chunk c ^ ((r >> 1) & 3) -- conflict-free for the 16x16x32 operand read at any
// first row AND for the epilogues' 16-byte stores of 8 consecutive rows (rpn_swz in respair.hip)
__device__ __forceinline__ int rc_swz(int row) { return ((row >> 1) & 3) << 4; }

}  // namespace

// VL: ragged batches (respair_wide.hip)
template <bool ACC, bool VL>
__global__ __launch_bounds__(RC_NT, 1) void reschain_kernel(const ResChainParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Xs = smem;
    char* const As = smem + RC_XBYTES;
    char* const W1s = As + RC_ABYTES;
    char* const W2s = W1s + RC_WBYTES;
    float* const Bs = reinterpret_cast<float*>(W2s + RC_WBYTES);      // [pair][conv][32]

    // MODE.FP16_OVFL (hwreg 1, bit 23): a conversion to fp16 that overflows gives +-65504 instead of +-inf -- the clamp
    // the pair kernels spend two VALU ops per element on
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int wm0 = wave * (RC_RT * 16);
    const int k = p.k, c = (k - 1) / 2;
    const int H = c * (p.dil[0] + p.dil[1] + p.dil[2] + 3);            // rows lost per side over the three pairs: 6 (k - 1) for dilations (1, 3, 5)
    const int Rout = RC_R - 2 * H;
    const int tiles_x = (p.L + Rout - 1) / Rout;
    const int total = VL ? p.total_tiles : tiles_x * p.B;
    auto tile_of = [&](int t, int& tb, int& tm0, int& tL) {            // tile -> clip, first output row, the clip's rows
        if constexpr (VL) { const SiVlTile v = si_vl_tile(p.lens, p.B, Rout, t); tb = v.b; tm0 = v.row0; tL = v.L; }
        else { tb = t / tiles_x; tm0 = (t - tb * tiles_x) * Rout; tL = p.L; }
    };
    const int wbytes = k * RC_C * RC_ROWB;

    // ---- weights of one convolution -> registers -> LDS (all taps; [tap][n][ci] is one contiguous run)
    u32x4 rw[RC_WSLOTS];
    auto issueW = [&](const unsigned short* w) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(w), 0, wbytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < RC_WSLOTS; ++i) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + i * RC_NT) * 16, 0, 0);
    };
    auto storeW = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < RC_WSLOTS; ++i) {
            const int q = tid + i * RC_NT;                             // 16-byte chunk of the run: row q / 4 (tap * 32 + n), chunk q % 4
            if (q * 16 < wbytes) {
                const int r = q >> 2, ch = q & 3;
                *reinterpret_cast<u32x4*>(dst + r * RC_ROWB + ((ch << 4) ^ rc_swz(r))) = rw[i];
            }
        }
    };
    // ---- the tile's rows of the input stream -> registers (rows outside the clip read as zero through the descriptor)
    u32x4 ry[RC_YSLOTS];
    auto issueY = [&](int t) {
        int tb, tm0, tL;
        tile_of(t, tb, tm0, tL);
        const int g0 = tm0 - H;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.y16 + (long)tb * p.L * RC_C), 0, tL * RC_C * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < RC_YSLOTS; ++i) {
            const int q = tid + i * RC_NT;
            const int g = g0 + (q >> 2);
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, g < 0 ? (int)0x80000000 : (g * RC_C + 8 * (q & 3)) * 2, 0, 0);
        }
    };

    // ---- once per workgroup: biases, zero margins of A, W1 of pair 0 in LDS, W2 of pair 0 and the first tile on their way
    if (tid < 6 * RC_C / 4) {
        const int which = tid / (RC_C / 4), c4 = (tid % (RC_C / 4)) * 4;   // which = pair * 2 + conv
        const float* src = (which & 1) ? p.b2[which >> 1] : p.b1[which >> 1];
        *reinterpret_cast<f32x4*>(Bs + which * RC_C + c4) = *reinterpret_cast<const f32x4*>(src + c4);
    }
    if (tid < 2 * RC_MARG * 4) {
        const int r = tid >> 2, ch = tid & 3;
        const int row = r < RC_MARG ? r : RC_R + r;                   // rows [0, MARG) and [MARG + R, MARG + R + MARG)
        *reinterpret_cast<u32x4*>(As + row * RC_ROWB + (ch << 4)) = u32x4{0u, 0u, 0u, 0u};
    }
    issueW(p.w1[0]);
    issueY(blockIdx.x);
    storeW(W1s);
    issueW(p.w2[0]);

    const int preW = r16 * RC_ROWB + (rc_swz(r16) ^ (kg << 4));        // + tap * 2048 + j * 1024: 16 rows keep the swizzle term

    f32x4 acc[RC_RT][2];                                               // [time tile i][channel tile j], transposed 16 x 16 tiles
    // the accumulators start from the bias of this lane's four consecutive channels (16 j + 4 kg ...): no add per element later
    auto init_acc = [&](const float* bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 4 * kg), b1v = *reinterpret_cast<const f32x4*>(bias + 16 + 4 * kg);
#pragma unroll
        for (int i = 0; i < RC_RT; ++i) { acc[i][0] = b0; acc[i][1] = b1v; }
    };
    // Operand addresses: linear byte offset of (row, k group) of row tile 0 plus the tap's row offset (a scalar), then the
    // swizzle as an XOR of address bits: bits 7-8 (row bits 1-2) -> bits 4-5.  Row tile i is 16 i rows = 1024 i bytes further:
    // above every bit the swizzle reads or writes, so the six fragment reads of a tap share ONE swizzled address and differ
    // in the instruction's immediate offset -- 4 VALU ops per tap instead of 24 (the VALU shares the SIMD's issue port with
    // the MFMAs; the compiler does not find this form from per-tile addresses).
    const int lin0 = (wm0 + r16 + RC_MARG) * RC_ROWB + (kg << 4);
    // lane (r16, kg) of accumulator tile (i, j): row wm0 + 16 i + r16, channels 16 j + 4 kg ... + 3 -> byte offset in a
    // swizzled 64-byte-row image whose row 0 is tile row -`shift`
    auto el_off = [&](int i, int j, int shift) {
        const int r = wm0 + 16 * i + r16 + shift;
        const int n = 16 * j + 4 * kg;
        return r * RC_ROWB + ((((n >> 3) << 4) ^ rc_swz(r)) + 8 * (kg & 1));
    };
    // one convolution over the tile: acc^T[row] += sum_tap W[tap] * A[row + (tap - c) * dd]^T; fragments double-buffered in
    // registers (the reads of tap + 1 are issued before the MFMAs of tap; clamped at the end: a harmless re-read)
    auto conv = [&](const char* Wc, int dd) {
        auto load = [&](f16x8 (&y)[RC_RT], f16x8 (&w)[2], int tap) {
#ifdef RC_ABLATE_READS                                                 // diagnostic build: results are garbage, timing only
            if (tap > 0) return;
#endif
            const int lin = lin0 + (tap - c) * dd * RC_ROWB;
            const char* ap = As + (lin ^ ((lin >> 3) & 0x30));
#pragma unroll
            for (int i = 0; i < RC_RT; ++i) y[i] = *reinterpret_cast<const f16x8*>(ap + i * (16 * RC_ROWB));
            const char* wp = Wc + tap * (RC_C * RC_ROWB) + preW;
#pragma unroll
            for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const f16x8*>(wp + j * (16 * RC_ROWB));
        };
        auto mma = [&](const f16x8 (&y)[RC_RT], const f16x8 (&w)[2]) {
#ifdef RC_ABLATE_HALF                                                  // diagnostic build: only one wave per SIMD issues MFMAs
            if (wave >= 4) return;
#endif
#ifdef RC_ABLATE_MFMA                                                  // diagnostic build: one MFMA per operand pair keeps the reads alive
#pragma unroll
            for (int i = 0; i < RC_RT; ++i) acc[i][0] += __builtin_bit_cast(f32x4, y[i]);
            acc[0][1] += __builtin_bit_cast(f32x4, w[0]) + __builtin_bit_cast(f32x4, w[1]);
            return;
#endif
#pragma unroll
            for (int i = 0; i < RC_RT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
        };
        f16x8 ya[RC_RT], wa[2], yb[RC_RT], wb[2];
        load(ya, wa, 0);
        for (int tap = 0; tap < k; tap += 2) {                         // k is odd: the last trip runs one tap
            load(yb, wb, tap + 1 < k ? tap + 1 : k - 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma(ya, wa);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            load(ya, wa, tap + 2 < k ? tap + 2 : k - 1);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < k) {
                __builtin_amdgcn_s_setprio(1);
                mma(yb, wb);
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    RC_TL_DECL
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int nxt = tile + (int)gridDim.x < total ? tile + (int)gridDim.x : tile;   // clamped: the loads stay unconditional
        int b, tm0, Lb;                                                // clip, first output row, the clip's rows
        tile_of(tile, b, tm0, Lb);
        const int g0 = tm0 - H;                                        // clip row of tile row 0
        const long seg = (long)b * p.L * RC_C;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out16 + seg, 0, Lb * RC_C * 2, 0x00020000);
        const bool edge = g0 < 0 || g0 + RC_R > Lb;                    // some rows of the tile lie outside the clip (workgroup-uniform)

        // ---- the tile (requested one tile ago, or at kernel entry) -> X raw, A leaky-ReLU(0.1) on the packed halves
#pragma unroll
        for (int i = 0; i < RC_YSLOTS; ++i) {
            const int q = tid + i * RC_NT;
            const int r = q >> 2, ch = q & 3;
            f16x8 h = __builtin_bit_cast(f16x8, ry[i]);
            *reinterpret_cast<f16x8*>(Xs + r * RC_ROWB + ((ch << 4) ^ rc_swz(r))) = h;
            h = __builtin_elementwise_max(h, h * (_Float16)0.1f);     // leaky-ReLU(0.1) = max(x, 0.1 x), packed
            const int ra = r + RC_MARG;
            *reinterpret_cast<f16x8*>(As + ra * RC_ROWB + ((ch << 4) ^ rc_swz(ra))) = h;
        }
        __syncthreads();
        RC_TL(0)

        // accumulate operand of the last pair, in the accumulator layout (8 bytes per lane); requested before the last c2
        u32x2 prev[ACC ? RC_RT : 1][2];
        // Epilogue stores.  In the accumulator layout a lane holds 4 channels (8 bytes) of one row and the 16 lanes of a
        // store group hold the SAME 8-byte column of 16 consecutive rows: on 64-byte rows that is 4 bank slots for 16
        // lanes, a 4-way conflict on every ds_write_b64 (measured: the epilogues were bound by it).  Lanes l and l + 16
        // hold the two halves of one 16-byte chunk, so for a PAIR of row tiles (i0, i1) they trade halves
        // (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second): afterwards a lane with
        // even k group owns chunk (2 j + kg / 2) of row tile i0, one with odd k group the same chunk of row tile i1 --
        // half as many stores, 16 bytes each.
        auto swap_halves = [&](u32x2& p0, u32x2& p1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const auto r = __builtin_amdgcn_permlane16_swap(p0[q], p1[q], false, false);
                p0[q] = r[0]; p1[q] = r[1];
            }
        };
        // byte offset of this lane's chunk after the trade, in an image whose row 0 is tile row -`shift`
        auto chunk_off = [&](int ip, int j, int shift) {
            const int r = wm0 + 16 * (2 * ip + (kg & 1)) + r16 + shift;
            return r * RC_ROWB + ((((2 * j + (kg >> 1)) << 4)) ^ rc_swz(r));
        };
        // c1 epilogue: t = lrelu(acc) (the bias is already in), zero outside the clip, fp16, over A.  The conversions
        // saturate (MODE.FP16_OVFL, set at kernel entry).  MASK only on tiles that reach past the clip.
        auto epi1 = [&](auto maskc) {
            constexpr bool MASK = decltype(maskc)::value;
#pragma unroll
            for (int ip = 0; ip < RC_RT / 2; ++ip) {
                float inside[2] = {1.f, 1.f};
                if constexpr (MASK) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) { const int g = g0 + wm0 + 16 * (2 * ip + u) + r16; inside[u] = (g >= 0 && g < Lb) ? 1.f : 0.f; }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x2 pk[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f16x4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = si_lrelu01(acc[2 * ip + u][j][e]);
                            if constexpr (MASK) v *= inside[u];
                            hv[e] = (_Float16)v;
                        }
                        pk[u] = __builtin_bit_cast(u32x2, hv);
                    }
                    swap_halves(pk[0], pk[1]);
                    *reinterpret_cast<u32x4*>(As + chunk_off(ip, j, RC_MARG)) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                }
            }
        };
        // c2 epilogue: x' = (acc + x) * alpha (+ previous), fp16, zero outside the clip; X <- x'; A <- lrelu(x') unless LAST.
        // All X reads of a half tile first: the compiler cannot move a read above an earlier X write.
        auto epi2 = [&](auto maskc, auto lastc) {
            constexpr bool MASK = decltype(maskc)::value, LAST = decltype(lastc)::value;
            constexpr int HP = (RC_RT / 2 + 1) / 2;                     // row-tile pairs per half: 2, then 1
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ip0 = h * HP, ip1 = h == 0 ? HP : RC_RT / 2;
                f16x4 xr[HP][2][2];
#pragma unroll
                for (int ip = ip0; ip < ip1; ++ip)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 2; ++j) xr[ip - ip0][u][j] = *reinterpret_cast<const f16x4*>(Xs + el_off(2 * ip + u, j, 0));
#pragma unroll
                for (int ip = ip0; ip < ip1; ++ip) {
                    _Float16 inside[2] = {(_Float16)1.f, (_Float16)1.f};
                    if constexpr (MASK) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) { const int g = g0 + wm0 + 16 * (2 * ip + u) + r16; inside[u] = (g >= 0 && g < Lb) ? (_Float16)1.f : (_Float16)0.f; }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        u32x2 pk[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int i = 2 * ip + u;
                            f16x4 xv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[i][j][e] + (float)xr[ip - ip0][u][j][e];
                                if constexpr (LAST) {
                                    v *= p.alpha;
                                    if constexpr (ACC) v += (float)__builtin_bit_cast(f16x4, prev[i][j])[e];
                                }
                                xv[e] = (_Float16)v;
                            }
                            if constexpr (MASK) xv *= inside[u];       // rows outside the clip stay zero for the next convolution
                            pk[u] = __builtin_bit_cast(u32x2, xv);
                        }
                        swap_halves(pk[0], pk[1]);
                        const u32x4 xc = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                        *reinterpret_cast<u32x4*>(Xs + chunk_off(ip, j, 0)) = xc;
                        if constexpr (!LAST) {
                            const f16x8 hx = __builtin_bit_cast(f16x8, xc);
                            *reinterpret_cast<f16x8*>(As + chunk_off(ip, j, RC_MARG)) = __builtin_elementwise_max(hx, hx * (_Float16)0.1f);
                        }
                    }
                }
            }
        };
        auto pair = [&](int pr, auto lastc) {
            constexpr bool LAST = decltype(lastc)::value;
            // ======== c1
            init_acc(Bs + (2 * pr) * RC_C);
            conv(W1s, p.dil[pr]);
            storeW(W2s);                                               // c2's weights of this pair (the previous c2 is long done)
            issueW(p.w1[LAST ? 0 : pr + 1]);
            __syncthreads();                                           // every wave has finished reading A
            RC_TL(1)
            if (edge) epi1(std::true_type{}); else epi1(std::false_type{});
            __syncthreads();                                           // t and W2 are visible
            RC_TL(2)
            // ======== c2
            if constexpr (LAST) {
                issueY(nxt);                                           // the next tile travels under the last convolution
                if constexpr (ACC) {
#pragma unroll
                    for (int i = 0; i < RC_RT; ++i) {
                        const int m = wm0 + 16 * i + r16;
                        const int g = g0 + m;
                        const bool keep = m >= H && m < RC_R - H && g < Lb;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            prev[i][j] = __builtin_amdgcn_raw_buffer_load_b64(orsrc, keep ? (g * RC_C + 16 * j + 4 * kg) * 2 : (int)0x80000000, 0, 0);
                    }
                }
            }
            init_acc(Bs + (2 * pr + 1) * RC_C);
            conv(W2s, 1);
            storeW(W1s);                                               // c1's weights of the next pair / of the next tile's pair 0
            issueW(p.w2[LAST ? 0 : pr + 1]);
            __syncthreads();                                           // every wave has finished reading t
            RC_TL(3)
            RC_TL_W0
            if (edge) epi2(std::true_type{}, lastc); else epi2(std::false_type{}, lastc);
            RC_TL_W1(6)
            RC_TL_W0
            __syncthreads();                                           // x' (and W1) are visible
            RC_TL_W1(7)
            RC_TL(4)
        };
#pragma unroll 1
        for (int pr = 0; pr < 2; ++pr) pair(pr, std::false_type{});
        pair(2, std::true_type{});

        // ---- output pass: rows [H, R - H) of X inside the clip, 16 bytes per lane, whole rows per 4 lanes
        for (int q = tid; q < Rout * 4; q += RC_NT) {
            const int r = H + (q >> 2), ch = q & 3;
            const int g = g0 + r;
            if (g < Lb) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(Xs + r * RC_ROWB + ((ch << 4) ^ rc_swz(r)));
                __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (g * RC_C + 8 * ch) * 2, 0, 0);
            }
        }
        __syncthreads();                                               // X is consumed: the next tile may stage into LDS
        RC_TL(5)
        RC_TL_TILE
    }
    RC_TL_FLUSH
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller launches the pairs one by one).
int si_launch_reschain(si_ctx* ctx, int C, const ResChainParams& p, hipStream_t st) {
    if (C != RC_C || p.k < 3 || p.k > RC_KMAX || (p.k & 1) == 0 || ((long)p.L + 2048) * C * 2 >= (1L << 31) || p.B <= 0 || p.L <= 0) return 1;
    for (int i = 0; i < 3; ++i) {
        if (p.dil[i] < 1 || (p.k - 1) / 2 * p.dil[i] > RC_MARG || !p.w1[i] || !p.w2[i] || !p.b1[i] || !p.b2[i]) return 1;
    }
    const int H = (p.k - 1) / 2 * (p.dil[0] + p.dil[1] + p.dil[2] + 3);
    if (2 * H > RC_R / 4) return 1;                                    // more than a quarter of the tile recomputed: not worth chaining
    if ((p.lens == nullptr) != (p.lens_host == nullptr)) return si_fail(ctx, SI_EINVAL, "reschain: ragged batches need the lengths on the device and on the host");
    const bool vl = p.lens != nullptr;
    auto kern = vl ? (p.accumulate ? reschain_kernel<true, true> : reschain_kernel<false, true>) : (p.accumulate ? reschain_kernel<true, false> : reschain_kernel<false, false>);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), RC_LDS)) return rc;
    const int Rout = RC_R - 2 * H;
    const int total = vl ? (int)si_vl_tiles(p.lens_host, p.B, Rout) : ((p.L + Rout - 1) / Rout) * p.B;
    if (total <= 0) return SI_OK;
    ResChainParams pk = p;
    pk.total_tiles = total;
    const int grid = std::min(total, si_num_cus(ctx));
    double rows = (double)p.B * p.L;
    if (vl) { rows = 0; for (int b = 0; b < p.B; ++b) rows += p.lens_host[b]; }
    const double elems = rows * C;
    si_prof_begin(ctx, p.accumulate ? "reschain_f16_c32_acc" : "reschain_f16_c32", 3 * 2.0 * 2.0 * elems * C * p.k, elems * (2.0 + 2.0 + (p.accumulate ? 2.0 : 0.0)) + 3 * 2.0 * 2.0 * p.k * C * C, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RC_NT), RC_LDS, st, pk);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
