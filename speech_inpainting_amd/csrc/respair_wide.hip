// respair_wide.hip -- one ResBlock1 step  y' = y + conv2(lrelu(conv1(lrelu(y))))  as ONE kernel for the WIDE vocoder
// stages (C = 128 / 256 channels) in the fp16 activation-stream mode (gfx950, wave64, MFMA).
//
// I_ea/hifi_gan/models.py:36-43 runs, per (resblock, dilation): xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt);
// x = xt + x; the MRF mean over the three resblocks is :112-118.  The wide stages carry 2/3 of the path's FLOPs.  As two
// tap-GEMM launches each convolution paid its own cold prologue and output burst per 256x128 tile and the pair's
// intermediate made a round trip through HBM / L2; here one 8-wave workgroup owns R1 rows x ALL C channels and keeps the
// intermediate in LDS:
//   prologue  the halo'd activation tile (R1 + (k-1)(d+... rows, all channels) is loaded ONCE, leaky-ReLU'd on the packed
//             halves and staged into LDS
//   phase 1   t[R1 rows] = lrelu(conv1(lrelu(y)) + b1), zero outside the clip (conv2's padding applies to t), rounded
//             to fp16 and written OVER the activation tile (dead once conv1 is done)
//   phase 2   acc[R1 rows] = conv2(t); rows >= R1 - (k-1) are recomputed by the neighbouring tile
//   epilogue  the fp32 accumulators go through an LDS image of the output tile (everything else is dead), then
//             out = (acc + y) * alpha (+ previous out) (the accumulators start from b2) is computed and stored row-contiguously: 16 bytes per lane,
//             whole 256 / 512-byte rows per wave instruction, residual / accumulate reads in the same shape
// Weight slabs ([C n][BKW ci] of one tap: 32 KB) stream L2 -> registers -> LDS through a double buffer, one workgroup
// barrier per slab; a slab feeds 32 (C = 128) / 16 (C = 256) MFMAs per wave, against 8 per barrier in the tap-GEMM.
//
// MFMA: v_mfma_f32_16x16x32_f16, 4 x 4 tiles per 64 x 64 wave tile.  Under a dense MFMA stream the chip holds ~1.8 GHz on
// this shape against ~1.45 GHz on 32x32x16 (tools/ubench/mfma_rate.hip: 1.82 vs 1.37-1.48 PFLOP/s sustained, LDS-fed), at
// the same LDS traffic per flop.  Orientation: D^T = W * Y^T (the WEIGHT rows are the A operand): a lane then holds one
// time row (column l & 15) and, in its 4 accumulator registers, four CONSECUTIVE channels -- so the intermediate is
// written to LDS with 8-byte stores and the output image with 16-byte stores, no shuffles.
//
// LDS images are unpadded; 16-byte chunk c of row r lives at chunk c ^ f(r) (swz16 below).  The XOR costs one VALU op per
// k-step and operand: pre = row base | ((f(r) ^ k group) << 4) of row / channel tile 0 per tap, address = pre ^ (k-step * 64),
// and tiles 1-3 (16 rows apart: same f) at immediate offsets of the same address register.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// (chunk XOR term of a row) << 4.  The 16x16x32 operand read of a ds_read_b128 lane group is 8 rows x chunk c and 8 other
// rows x chunk c ^ 1 (lane l: row l & 15, k group l >> 4), so the term leaves chunk bit 0 alone -- the two halves of a group
// can then never meet -- and spreads 8 consecutive rows over the 8 slot pairs: conflict-free for ANY first row (a tap is
// an arbitrary row offset).  (f = ((r & 7) << 1) ^ ((r >> 2) & 1) would also make the phase-1 epilogue's 16-byte stores of 8
// consecutive rows conflict-free -- they are 2-way conflicted under this f -- as the 64- / 128-byte-row swizzles of
// respair.hip and reschain.hip do; measured here: no gain, one more VALU op per address, not adopted.)
template <int ROWB>
__device__ __forceinline__ int swz16(int row) {
    if constexpr (ROWB >= 256) return (row & 7) << 5;
    else return ((row >> 1) & 3) << 5;                                // 128-byte rows: two rows per 256-byte bank period
}

// Diagnostic build only (make timeline; tools/exp_vocoder_only.py): per-phase wall-clock totals of every workgroup's
// wave 0, in 100 MHz ticks.  [width 0 = C128, 1 = C256][phase]: 0 tile staging, 1 conv-1 slabs, 2 phase-1 epilogue,
// 3 conv-2 slabs, 4 accumulators -> output image, 5 output pass, 6 tiles, 7 workgroups, 8 / 9: of phase 4, wave 0's wait at the
// barrier behind the last slab / its image stores.
#ifdef RPW_TIMELINE
__device__ unsigned long long rpw_tl[2][10];
#define RPW_TL_DECL unsigned long long tl_t = wall_clock64(); unsigned long long tl_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tl_tiles = 0; unsigned long long tl_w = 0, tl_sub[2] = {0, 0};
#define RPW_TL_W0 tl_w = wall_clock64();
#define RPW_TL_W1(q) tl_sub[q] += wall_clock64() - tl_w;
#define RPW_TL(ph) { const unsigned long long n_ = wall_clock64(); tl_acc[ph] += n_ - tl_t; tl_t = n_; }
#define RPW_TL_TILE ++tl_tiles;
#define RPW_TL_FLUSH if (threadIdx.x == 0) { for (int q_ = 0; q_ < 6; ++q_) atomicAdd(&rpw_tl[C == 256][q_], tl_acc[q_]); atomicAdd(&rpw_tl[C == 256][6], tl_tiles); atomicAdd(&rpw_tl[C == 256][7], 1ull); atomicAdd(&rpw_tl[C == 256][8], tl_sub[0]); atomicAdd(&rpw_tl[C == 256][9], tl_sub[1]); }
extern "C" int si_debug_rpw_timeline(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rpw_tl), sizeof(rpw_tl)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[2][10] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(rpw_tl), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define RPW_TL_DECL
#define RPW_TL_W0
#define RPW_TL_W1(q)
#define RPW_TL(ph)
#define RPW_TL_TILE
#define RPW_TL_FLUSH
#endif

constexpr int RPW_HALO = 50;                                          // (k - 1) * dil <= 50: k = 11, dil = 5

// ACC: the launch adds into the previous contents of out16 (last pair of the 2nd / 3rd resblock).  Without it the
// registers of the accumulate rows are free and the NEXT tile's activation rows are prefetched under the last slab.
// VL: ragged batches -- clip b holds p.lens[b] rows (at stride p.L); tiles are numbered clip by clip without gaps (si_vl_tile).
template <int C, int R1, int WARPS_M, int WARPS_N, int BKW, bool ACC, bool VL>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, 2) void respair_wide_kernel(const ResPairParams p) {
    static_assert((WARPS_M * WARPS_N == 8 || WARPS_M * WARPS_N == 4) && R1 == WARPS_M * 64 && C == WARPS_N * 64, "64 x 64 wave tiles");
    static_assert(BKW == 128 || BKW == 64, "weight-slab depth");
    constexpr int NT = 64 * WARPS_M * WARPS_N;                         // 512: one workgroup per CU; 256: two (two waves per SIMD either way)
    constexpr int ROWBY = C * 2;                                       // bytes per activation / intermediate row
    constexpr int ROWBW = BKW * 2;                                     // bytes per weight-slab row
    constexpr int ROWBO = C * 4;                                       // bytes per fp32 output-image row
    constexpr int CPRY = C / 8, CPRW = BKW / 8;                        // 16-byte chunks per row
    constexpr int NCH = C / BKW;                                       // weight slabs per tap
    constexpr int KS = BKW / 32;                                       // MFMA k-steps (K = 32) per slab
    constexpr int YBYTES = (R1 + RPW_HALO) * ROWBY;
    constexpr int WBYTES = C * ROWBW;
    constexpr int WSLOTS = C * CPRW / NT;
    constexpr int YSLOTS = ((R1 + RPW_HALO) * CPRY + NT - 1) / NT;
    constexpr int YRPP = NT / CPRY, WRPP = NT / CPRW;                  // rows per staging pass
    static_assert(C * CPRW % NT == 0 && NT % CPRY == 0 && NT % CPRW == 0, "staging maps");
    static_assert(R1 * ROWBO <= YBYTES + 2 * WBYTES, "the output image reuses the operand region (not the biases behind it)");

    // MODE.FP16_OVFL (hwreg 1, bit 23): a conversion to fp16 that overflows gives +-65504 instead of +-inf, i.e. the
    // clamp before every rounding comes with the conversion
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ys = smem;                                             // [R1 + 50][C] fp16: lrelu(y), later t (rows < R1)
    char* const Ws = smem + YBYTES;                                    // [2][C][BKW] fp16 weight slabs
    float* const Bs = reinterpret_cast<float*>(smem + YBYTES + 2 * WBYTES);   // [2][C] fp32: b1, b2 (outside the output image)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;                        // operand row / k group of this lane (also: C column / row group)
    const int wm0 = (wave / WARPS_N) * 64, wn0 = (wave % WARPS_N) * 64;
    const int k = p.k, d = p.dil;
    const int p1 = d * (k - 1) / 2, p2 = (k - 1) / 2;
    const int BMo = R1 - (k - 1);
    const int R0 = R1 + (k - 1) * d;
    // ---- persistent workgroups: tile = (clip, row block); a workgroup walks tiles blockIdx.x, + gridDim.x, ...
    const int tiles_x = (p.L + BMo - 1) / BMo;
    const int total = VL ? p.total_tiles : tiles_x * p.B;
    auto tile_of = [&](int t, int& tb, int& tm0, int& tL) {            // tile -> clip, first output row, the clip's rows
        if constexpr (VL) { const SiVlTile v = si_vl_tile(p.lens, p.B, BMo, t); tb = v.b; tm0 = v.row0; tL = v.L; }
        else { tb = t / tiles_x; tm0 = (t - tb * tiles_x) * BMo; tL = p.L; }
    };
    const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w1), 0, k * C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w2), 0, k * C * C * 2, 0x00020000);

    // ---- weight slabs: slab s = (conv, tap, chunk) in that order; conv 1 first
    const int NS1 = k * NCH, NS = 2 * NS1;
    const int wc = tid % CPRW, wr0 = tid / CPRW;
    u32x4 rw[WSLOTS];
    auto issueW1 = [&](int s, int i) {                                 // slot i (rows wr0 + i * WRPP) of slab s
        const bool second = s >= NS1;
        const int q = second ? s - NS1 : s;
        const int tap = q / NCH, ch = q - tap * NCH;
        // readfirstlane: the slab offset is wave-uniform, but the compiler cannot always prove it and would wrap every
        // load in a waterfall loop
        const int soff = __builtin_amdgcn_readfirstlane((tap * C * C + ch * BKW) * 2);
        const int n = wr0 + i * WRPP;
        rw[i] = __builtin_amdgcn_raw_buffer_load_b128(second ? w2rsrc : w1rsrc, (n * C + 8 * wc) * 2, soff, 0);
    };
    auto issueW = [&](int s) {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) issueW1(s, i);
    };
    auto storeW1 = [&](char* dst, int i) {
        const int n = wr0 + i * WRPP;
        *reinterpret_cast<u32x4*>(dst + n * ROWBW + ((wc << 4) ^ swz16<ROWBW>(n))) = rw[i];
    };
    auto storeW = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) storeW1(dst, i);
    };
    // the halo'd activation rows of a tile -> registers (rows outside the clip read as zero through the clip's descriptor)
    const int yc = tid % CPRY, yr0 = tid / CPRY;
    u32x4 ry[YSLOTS];
    auto issueY = [&](int t) {
        int tb, tm0, tL;
        tile_of(t, tb, tm0, tL);
        const int trow0 = tm0 - p2 - p1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.y16 + (long)tb * p.L * C), 0, tL * C * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < YSLOTS; ++i)
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((trow0 + yr0 + i * YRPP) * C + 8 * yc) * 2, 0, 0);
    };

    if (tid < C / 2) {                                                 // biases -> LDS: the epilogues read them per lane
        const int which = tid / (C / 4), c4 = (tid % (C / 4)) * 4;
        *reinterpret_cast<f32x4*>(Bs + which * C + c4) = *reinterpret_cast<const f32x4*>((which ? p.b2 : p.b1) + c4);
    }
    // channel tile j of a slab is 16 j rows further: 16 rows keep the swizzle term, so the four fragments of a k-step share one
    // address and differ in the instruction's immediate offset (the VALU shares the SIMD's issue port with the MFMAs)
    const int preW = (wn0 + r16) * ROWBW + (swz16<ROWBW>(wn0 + r16) ^ (kg << 4));
    issueW(0);
    issueY(blockIdx.x);
    RPW_TL_DECL

  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int nxt = tile + (int)gridDim.x < total ? tile + (int)gridDim.x : tile;   // clamped: the loads below stay unconditional
    int b, m0, Lb;                                                     // clip, first output row of this tile, the clip's rows
    tile_of(tile, b, m0, Lb);
    const int t_row0 = m0 - p2;                                        // clip row of intermediate row 0
    const long seg = (long)b * p.L * C;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.y16 + seg), 0, Lb * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out16 + seg, 0, Lb * C * 2, 0x00020000);

    // ---- slab 0 and the activation tile (both requested one tile ago, or at kernel entry) -> LDS: raw fp16 -> leaky-ReLU(0.1)
    //      on the packed halves
    storeW(Ws);
    issueW(1);                                                         // stays in registers until slab 0 starts (NS >= 6)
#pragma unroll
    for (int i = 0; i < YSLOTS; ++i) {
        const int r = yr0 + i * YRPP;
        if ((i + 1) * YRPP <= R1 || r < R0) {                           // rows < R1 always exist: no branch around their loads
            f16x8 h = __builtin_bit_cast(f16x8, ry[i]);
            h = __builtin_elementwise_max(h, h * (_Float16)0.1f);     // leaky-ReLU(0.1) = max(x, 0.1 x), packed
            *reinterpret_cast<f16x8*>(Ys + r * ROWBY + ((yc << 4) ^ swz16<ROWBY>(r))) = h;
        }
    }
    __syncthreads();
    RPW_TL(0)

    f32x4 acc[4][4];                                                   // [time tile i][channel tile j], transposed 16 x 16 tiles
    // the accumulators start from the bias of this lane's four consecutive channels: no add per element in the epilogues
    auto init_acc = [&](const float* bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + wn0 + 16 * j + 4 * kg);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = bv;
        }
    };
    init_acc(Bs);
    // one slab: acc^T += W[slab] * A[rows + roff][chunk columns]^T.  Fragments are double-buffered in registers: the
    // reads of k-step ks + 1 are issued before the MFMAs of k-step ks, so an MFMA never waits on a read issued just
    // ahead of it.  `hook(ks)` runs behind the MFMAs of k-step ks: the weight-slab stores and the next slab's loads are
    // spread over the k-steps there instead of forming a 32 KB ds_write burst at the slab boundary.
    auto compute = [&](int roff, int cb16, const char* Wc, auto&& hook) {
        const int r0 = wm0 + r16 + roff;                               // row tile i: 16 i rows further, same swizzle term
        const int preY = r0 * ROWBY + (swz16<ROWBY>(r0) ^ (kg << 4));
        auto load = [&](f16x8 (&y)[4], f16x8 (&w)[4], int ks) {
            const char* yp = Ys + (preY ^ (cb16 + ks * 64));
            const char* wp = Wc + (preW ^ (ks * 64));
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const f16x8*>(yp + i * (16 * ROWBY));
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f16x8*>(wp + j * (16 * ROWBW));
        };
        auto mma = [&](const f16x8 (&y)[4], const f16x8 (&w)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
        };
        // sched_barrier(0): left alone, the scheduler folds the two register sets back into one and puts each read right
        // behind the MFMA that frees its register, a few cycles ahead of its use
        f16x8 ya[4], wa[4], yb[4], wb[4];
        load(ya, wa, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            load(yb, wb, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma(ya, wa);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            hook(ks);
            if (ks + 2 < KS) load(ya, wa, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma(yb, wb);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            hook(ks + 1);
        }
    };

    int tap = 0, ch = 0;                                               // of slab s within its convolution
    for (int s = 0; s < NS - 1; ++s) {
        const bool second = s >= NS1;
        char* const Wn = Ws + ((s + 1) & 1) * WBYTES;
        static_assert(WSLOTS % KS == 0, "the slab's stores are spread evenly over its k-steps");
        compute(second ? tap : tap * d, ch * (BKW * 2), Ws + (s & 1) * WBYTES, [&](int ks) {
            // slot by slot: slab s + 1 leaves its register for LDS and the same slot of slab s + 2 is requested into it at
            // once, so every load has a whole slab of MFMAs to land (steady state: vmcnt(WSLOTS - 1) at each store).
            // Unconditional (clamped to the last slab): a conditional load drains vmcnt at the join.
            const int s2 = s + 2 < NS ? s + 2 : NS - 1;
#pragma unroll
            for (int i = 0; i < WSLOTS / KS; ++i) {
                storeW1(Wn, ks * (WSLOTS / KS) + i);
                issueW1(s2, ks * (WSLOTS / KS) + i);
            }
        });
        if (s == NS1 - 1) {
            __syncthreads();                                           // every wave has finished reading the activation tile
            RPW_TL(1)
            // ---- phase-1 epilogue: leaky-ReLU (the bias is in the accumulators), zero outside the clip, fp16, over the activation
            //      tile.  In the accumulator layout the 16 lanes of a store group hold one 8-byte column of 16 consecutive rows:
            //      a 4-way bank conflict per ds_write_b64 on these rows.  Lanes l and l + 16 hold the two halves of a 16-byte
            //      chunk, so for a pair of row tiles they trade halves (v_permlane16_swap: odd 16-lane rows of the first operand
            //      <-> even rows of the second): a lane with even k group then owns chunk (n / 8) of row tile 2 ip, one with
            //      odd k group the same chunk of row tile 2 ip + 1 -- half as many stores, 16 bytes each (reschain.hip).
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                float inside[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int grow = t_row0 + wm0 + 16 * (2 * ip + u) + r16;
                    inside[u] = (grow >= 0 && grow < Lb) ? 1.f : 0.f;         // as a factor: no branch per element
                }
                const int ms = wm0 + 16 * (2 * ip + (kg & 1)) + r16;           // the row this lane stores after the trade
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x2 pk[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f16x4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(si_lrelu01(acc[2 * ip + u][j][e]) * inside[u]);   // saturating (MODE.FP16_OVFL)
                        pk[u] = __builtin_bit_cast(u32x2, hv);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const auto r = __builtin_amdgcn_permlane16_swap(pk[0][q], pk[1][q], false, false);
                        pk[0][q] = r[0]; pk[1][q] = r[1];
                    }
                    const int ch = ((wn0 + 16 * j) >> 3) + (kg >> 1);          // 16-byte chunk (8 channels) of the row
                    *reinterpret_cast<u32x4*>(Ys + ms * ROWBY + ((ch << 4) ^ swz16<ROWBY>(ms))) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                }
            }
            init_acc(Bs + C);
            RPW_TL(2)
        }
        if (++ch == NCH) { ch = 0; if (++tap == k) tap = 0; }
        __syncthreads();                                               // slab s + 1 (and, after phase 1, the intermediate) is visible
    }
    // ---- last slab (tap k - 1, last chunk of conv 2).  Without an accumulate operand the next tile's activation rows are
    //      requested first, so that they travel under its MFMAs.
    if constexpr (!ACC) issueY(nxt);
    __builtin_amdgcn_sched_barrier(0);
    compute(tap, ch * (BKW * 2), Ws + ((NS - 1) & 1) * WBYTES, [](int) {});
    RPW_TL(3)
    // the residual / accumulate rows of the output pass travel under the output image's LDS round trip: a lane owns 8
    // consecutive channels of OPASS output rows
    constexpr int ORPP = NT / CPRY;                                    // output rows per pass
    constexpr int OPASS = R1 / ORPP;
    const int c8 = tid % CPRY, or0 = tid / CPRY;
    u32x4 res[OPASS], prev[ACC ? OPASS : 1];
    int goff[OPASS];
#pragma unroll
    for (int it = 0; it < OPASS; ++it) {
        const int o = or0 + it * ORPP;
        const int grow = m0 + o;
        goff[it] = (o < BMo && grow < Lb) ? (grow * C + 8 * c8) * 2 : (int)0x80000000;
        res[it] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, goff[it], 0, 0);
        if constexpr (ACC) prev[it] = __builtin_amdgcn_raw_buffer_load_b128(orsrc, goff[it], 0, 0);
    }
    RPW_TL_W0
    __syncthreads();                                                   // every wave is done with the operand tiles
    RPW_TL_W1(0)
    RPW_TL_W0

    // ---- final epilogue: accumulators -> fp32 image of the output tile in LDS -> row-contiguous residual add + store
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = wm0 + 16 * i + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = ((wn0 + 16 * j) >> 2) + kg;                 // 16-byte chunk (4 channels) of the output row
            *reinterpret_cast<f32x4*>(smem + m * ROWBO + ((co ^ (m & 15)) << 4)) = acc[i][j];
        }
    }
    RPW_TL_W1(1)
    __syncthreads();
    RPW_TL(4)
    issueW(0);                                                         // slab 0 of the next tile lands during the output pass
    {
        const bool oact = p.out_slope != 0.f && p.out_slope != 1.f;
#pragma unroll
        for (int it = 0; it < OPASS; ++it) {
            const int o = or0 + it * ORPP;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + o * ROWBO + (((2 * c8) ^ (o & 15)) << 4));
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + o * ROWBO + (((2 * c8 + 1) ^ (o & 15)) << 4));
            const f16x8 rh = __builtin_bit_cast(f16x8, res[it]);
            f16x8 ph = {};
            if constexpr (ACC) ph = __builtin_bit_cast(f16x8, prev[it]);
            f16x8 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = e < 4 ? a0[e] : a1[e - 4];
                float v = (a + (float)rh[e]) * p.alpha;
                if constexpr (ACC) v += (float)ph[e];
                if (oact) v = v > 0.f ? v : v * p.out_slope;           // the consumer's leaky-ReLU (uniform branch; see ResPairParams)
                out[e] = (_Float16)v;                                  // saturating (MODE.FP16_OVFL)
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, out), orsrc, goff[it], 0, 0);
        }
    }
    if constexpr (ACC) issueY(nxt);                                    // (no registers to spare earlier in this variant)
    __syncthreads();                                                   // the output image is consumed: the next tile may stage into LDS
    RPW_TL(5)
    RPW_TL_TILE
  }
  RPW_TL_FLUSH
}

template <int C, int R1, int WARPS_M, int WARPS_N, int BKW>
static int respair_wide_launch(si_ctx* ctx, const ResPairParams& p0, hipStream_t st) {
    const ResPairParams& p = p0;
    const int BMo = R1 - (p.k - 1);
    const size_t lds = (size_t)(R1 + RPW_HALO) * C * 2 + 2 * (size_t)C * BKW * 2 + 2 * (size_t)C * 4;
    const bool vl = p0.lens != nullptr;
    auto kern = vl ? (p.accumulate ? respair_wide_kernel<C, R1, WARPS_M, WARPS_N, BKW, true, true> : respair_wide_kernel<C, R1, WARPS_M, WARPS_N, BKW, false, true>)
                   : (p.accumulate ? respair_wide_kernel<C, R1, WARPS_M, WARPS_N, BKW, true, false> : respair_wide_kernel<C, R1, WARPS_M, WARPS_N, BKW, false, false>);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    // one persistent workgroup per CU (144-157 KB of LDS each) walking tiles blockIdx.x + i * gridDim.x
    const int total = vl ? (int)si_vl_tiles(p.lens_host, p.B, BMo) : ((p.L + BMo - 1) / BMo) * p.B;
    if (total <= 0) return SI_OK;
    ResPairParams pk = p0;
    pk.total_tiles = total;
    const int grid = std::min(total, si_num_cus(ctx) * (WARPS_M * WARPS_N == 4 ? 2 : 1));
    char name[48];
    snprintf(name, sizeof(name), p.accumulate ? "respair_f16_c%d_acc" : "respair_f16_c%d", C);   // one family per instantiation
    double rows = (double)p.B * p.L;
    if (vl) { rows = 0; for (int b = 0; b < p.B; ++b) rows += p.lens_host[b]; }
    const double elems = rows * C;
    si_prof_begin(ctx, name, 2.0 * 2.0 * elems * C * p.k, elems * (2.0 + 2.0 + (p.accumulate ? 2.0 : 0.0)) + 2.0 * 2.0 * p.k * C * C, st);   // y read once (it is also the residual), out written [, previous out read]
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WARPS_M * WARPS_N), lds, st, pk);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller launches the two convolutions).
// Shape limits (k odd in 3..11, (k - 1) * dil <= 50, 32-bit in-clip byte offsets) are checked by si_launch_respair.
int si_launch_respair_wide(si_ctx* ctx, int C, const ResPairParams& p, hipStream_t st) {
    // C = 128: one 8-wave workgroup per CU on 256 rows with whole-tap weight slabs.  (Measured against two 4-wave
    // workgroups per CU on 128 rows / half-tap slabs, which overlap one workgroup's tile load and output pass with the
    // other's MFMAs: 206 / 368 / 515 us per launch for k = 3 / 7 / 11 against 204 / 370 / 576 -- the second form streams
    // every weight slab twice as often per flop, and weight staging is the loop's largest overhead.  Measured again on the
    // persistent form with the slot-by-slot weight prefetch: the two-workgroup form is 10 % slower over the family.)
    if (C == 128) return respair_wide_launch<128, 256, 4, 2, 128>(ctx, p, st);
    if (C == 256) return respair_wide_launch<256, 128, 2, 4, 64>(ctx, p, st);
    return 1;
}
