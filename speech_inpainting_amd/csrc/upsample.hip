// upsample.hip -- the two LATE transposed convolutions of the generator (I_ea/hifi_gan/models.py:91-95,110-111: x = lrelu(x, 0.1);
// x = ups[i](x); ConvTranspose1d(C -> C/2, k = 4, stride 2, padding 1) for the 128- and 64-channel inputs) on the fp16 activation
// stream, as one streaming kernel each (gfx950, wave64, v_mfma_f32_16x16x32_f16).
//
// A transposed convolution with k = q * stride is a plain GEMM on overlapping input rows (api.hip packs the weights that way):
//     out_full[m][n] = sum_tap sum_ci  x[m - tap][ci] * W[tap][n][ci],   n = phase * Cout + co,  out[t][co] = out_full flat [t * Cout + co + pad * Cout]
// i.e. row m of the GEMM is a CONTIGUOUS run of `stride * Cout` output elements.  With N = 128 / 64 and K = 256 / 128 these two
// layers have 64 / 32 MACs per byte they move: HBM-bound (270 / 360 MB per 32 clips), and the generic tap-GEMM ran them at 2.1 / 3.0
// TB/s -- every 256-row tile a cold start: weights re-fetched, the input tile loaded, waited for, used, the output stored.  Here:
//   * persistent workgroups; the WHOLE weight matrix is staged into LDS once per workgroup;
//   * the input rows of the NEXT tile are requested (all at once, into registers) before the current tile is computed, so a tile's
//     HBM round trip hides behind the previous tile's MFMAs and stores;
//   * leaky-ReLU(0.1) on the packed halves while staging (max(x, 0.1 x)), LDS rows padded by 32 bytes instead of swizzled
//     (conflict-free operand reads, purely additive addresses: respair.hip), the bias in the accumulators' initial value;
//   * the epilogue trades halves between lanes l and l + 16 (v_permlane16_swap: 16 bytes per lane) and stores straight from the
//     accumulators (128 channels) or through an LDS image of the output tile, row-contiguous (64 channels).
// Arithmetic: the fp16 vocoder mode's -- fp16 operands (the stream is stored that way; the weights are rounded once at load),
// fp32 accumulation tap 0 first, one saturating rounding of the result (MODE.FP16_OVFL).
#include <algorithm>
#include <cstdio>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int UP_RT = 256;                                            // GEMM rows (= input rows) per tile
constexpr int UP_TAPS = 2;
}  // namespace

// CIN input channels, N = stride * Cout GEMM columns (N == CIN for stride 2).  Waves: 4 along M x N / 64 along N, 64 x 64 tiles.
// VL: ragged batches -- clip b holds p.lens_lin[b] input rows, p.lens_m[b] GEMM rows, p.lens_lin[b] * N output elements (common.h)
template <int CIN, int N, bool VL>
__global__ __launch_bounds__(64 * 4 * (N / 64), 2) void upsample_stream_kernel(const UpsampleParams p) {
    constexpr int WN = N / 64, NT = 64 * 4 * WN;
    constexpr int ROWB = CIN * 2 + 32;                                 // padded LDS row (activations and weights)
    constexpr int AROWS = UP_RT + UP_TAPS - 1;
    constexpr int ABYTES = AROWS * ROWB;
    constexpr int CPR = CIN / 8;                                       // 16-byte chunks per row
    constexpr int ASLOTS = (AROWS * CPR + NT - 1) / NT;
    constexpr int KS = CIN / 32;                                       // k-steps per tap
    static_assert(N % 64 == 0 && CIN % 32 == 0, "64-column wave tiles, 32-channel k-steps");
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);           // MODE.FP16_OVFL: conversions to fp16 saturate
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;                                             // [AROWS][ROWB]: lrelu(x) rows m0 - 1 ... m0 + 255
    char* const Ws = smem + ABYTES;                                    // [TAPS][N][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * 64;
    const int tiles_x = (p.M + UP_RT - 1) / UP_RT, total = VL ? p.total_tiles : tiles_x * p.B;
    // tile -> clip, first GEMM row, the clip's GEMM rows / input rows
    auto tile_of = [&](int t, int& tb, int& tm0, int& tM, int& tLin) {
        if constexpr (VL) { const SiVlTile v = si_vl_tile(p.lens_m, p.B, UP_RT, t); tb = v.b; tm0 = v.row0; tM = v.L; tLin = p.lens_lin[v.b]; }
        else { tb = t / tiles_x; tm0 = (t - tb * tiles_x) * UP_RT; tM = p.M; tLin = p.Lin; }
    };

    // ---- the input rows of a tile -> registers (rows before / after the clip read as zero through the clip's descriptor)
    // TWO tiles ahead (two register sets; one tile ahead measured the same: the layers run at 5.3 / 3.2 TB/s of mixed reads and writes)
    u32x4 ra[2][ASLOTS];
    auto issueA = [&](auto setc, int tile) {
        constexpr int SET = decltype(setc)::value;
        if (tile >= total) tile = total - 1;                           // clamped: the loads stay unconditional
        int tb, m0, tM, tLin;
        tile_of(tile, tb, m0, tM, tLin);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x16 + (long)tb * p.Lin * CIN), 0, tLin * CIN * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int q = tid + i * NT;
            const int r = q / CPR, c = q - r * CPR;
            const int xr = m0 - (UP_TAPS - 1) + r;
            ra[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (xr < 0 || r >= AROWS) ? (int)0x80000000 : (xr * CIN + 8 * c) * 2, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    issueA(S0{}, blockIdx.x);
    issueA(S1{}, blockIdx.x + gridDim.x);
    // ---- the weights, once: [tap][n][ci] fp16 -> padded rows
    for (int q = tid; q < UP_TAPS * N * CPR; q += NT) {
        const int r = q / CPR, c = q - r * CPR;
        *reinterpret_cast<u32x4*>(Ws + r * ROWB + c * 16) = *reinterpret_cast<const u32x4*>(p.w + (long)r * CIN + 8 * c);
    }
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + wn0 + 16 * j + 4 * kg);

    auto do_tile = [&](auto setc, int tile) {
        constexpr int SET = decltype(setc)::value;
        int b, m0, Mb, Linb;
        tile_of(tile, b, m0, Mb, Linb);
        const long oelems = VL ? (long)Linb * N : p.o_clip_elems;       // output elements of this clip
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int q = tid + i * NT;
            const int r = q / CPR, c = q - r * CPR;
            f16x8 h = __builtin_bit_cast(f16x8, ra[SET][i]);
            h = __builtin_elementwise_max(h, h * (_Float16)0.1f);     // leaky-ReLU(0.1) = max(x, 0.1 x), packed
            if (r < AROWS) *reinterpret_cast<f16x8*>(As + r * ROWB + c * 16) = h;
        }
        __syncthreads();
        issueA(setc, tile + 2 * (int)gridDim.x);

        f32x4 acc[4][4];                                               // [row tile i][column tile j], transposed 16 x 16 tiles (D^T = W . A^T)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = bv[j];
        // GEMM row ml, tap q reads tile row ml + (TAPS - 1) - q
        const char* ap = As + (wm0 + r16 + UP_TAPS - 1) * ROWB + (kg << 4);
        const char* wp = Ws + (wn0 + r16) * ROWB + (kg << 4);
#pragma unroll
        for (int tap = 0; tap < UP_TAPS; ++tap)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 y[4], w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const f16x8*>(ap + (16 * i - tap) * ROWB + ks * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f16x8*>(wp + (tap * N + 16 * j) * ROWB + ks * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
            }

        // ---- epilogue: fp16, lanes l and l + 16 trade halves (v_permlane16_swap) so that every lane owns a 16-byte chunk.  Element
        //      (m, n) of the GEMM is element m * N + n - ooff of the clip's output (a whole 8-element chunk is inside or outside:
        //      N, ooff and the clip size are multiples of 8).
        //      IMAGE (the 64-channel layer, two workgroups per CU): through an LDS image of the output tile (over the input tile,
        //      dead now), so that a global store instruction writes 1 KB of contiguous output: 76 -> 67 us.  The 128-channel layer
        //      (one workgroup per CU: nothing overlaps the two extra barriers) stores straight from the accumulators, 64 contiguous
        //      bytes per 4 lanes: 84 us against 100 through the image.
        constexpr bool IMAGE = N == 64;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out16 + (long)b * p.o_clip_stride, 0, (int)(oelems * 2), 0x00020000);
        if constexpr (IMAGE) __syncthreads();                          // every wave has finished reading the input tile
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int ms = wm0 + 16 * (2 * ip + (kg & 1)) + r16;       // the tile row this lane stores after the trade
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x2 pk[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) pk[u] = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[2 * ip + u][j], f16x4));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const auto r = __builtin_amdgcn_permlane16_swap(pk[0][q], pk[1][q], false, false);
                    pk[0][q] = r[0]; pk[1][q] = r[1];
                }
                const u32x4 v = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                const int n = wn0 + 16 * j + 8 * (kg >> 1);
                if constexpr (IMAGE) {
                    *reinterpret_cast<u32x4*>(As + ms * ROWB + n * 2) = v;
                } else {
                    const long e = (long)(m0 + ms) * N + n - p.ooff;
                    const int off = (m0 + ms < Mb && e >= 0 && e < oelems) ? (int)(e * 2) : (int)0x80000000;
                    __builtin_amdgcn_raw_buffer_store_b128(v, ors, off, 0, 0);
                }
            }
        }
        if constexpr (IMAGE) {
            __syncthreads();
            constexpr int OCPR = N / 8, OSLOTS = UP_RT * OCPR / NT;
            static_assert(UP_RT * OCPR % NT == 0, "output pass map");
#pragma unroll
            for (int i = 0; i < OSLOTS; ++i) {
                const int q = tid + i * NT;
                const int r = q / OCPR, c = q - r * OCPR;
                const u32x4 v = *reinterpret_cast<const u32x4*>(As + r * ROWB + c * 16);
                const long e = (long)(m0 + r) * N + 8 * c - p.ooff;
                const int off = (m0 + r < Mb && e >= 0 && e < oelems) ? (int)(e * 2) : (int)0x80000000;
                __builtin_amdgcn_raw_buffer_store_b128(v, ors, off, 0, 0);
            }
        }
        __syncthreads();                                               // the tile is consumed: the next one may be staged
    };
    for (int tile = blockIdx.x; tile < total; tile += 2 * gridDim.x) {
        do_tile(S0{}, tile);
        if (tile + (int)gridDim.x < total) do_tile(S1{}, tile + gridDim.x);
    }
}

template <int CIN, int N>
static int upsample_launch(si_ctx* ctx, const UpsampleParams& p, hipStream_t st) {
    constexpr int WN = N / 64;
    const size_t lds = (size_t)(UP_RT + UP_TAPS - 1 + UP_TAPS * N) * (CIN * 2 + 32);
    const bool vl = p.lens_m != nullptr;
    auto kern = vl ? upsample_stream_kernel<CIN, N, true> : upsample_stream_kernel<CIN, N, false>;
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    const int total = vl ? (int)si_vl_tiles(p.lens_m_host, p.B, UP_RT) : ((p.M + UP_RT - 1) / UP_RT) * p.B;
    if (total <= 0) return SI_OK;
    UpsampleParams pk = p;
    pk.total_tiles = total;
    const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
    const int grid = std::min(total, si_num_cus(ctx) * per_cu);
    char name[40];
    snprintf(name, sizeof(name), "upsample_f16_c%d", CIN);
    double rows = (double)p.B * p.Lin;                                 // input rows that exist (lens_m = input rows + 1 for k 4 / stride 2)
    if (vl) { rows = 0; for (int b = 0; b < p.B; ++b) rows += p.lens_m_host[b] > 0 ? p.lens_m_host[b] - 1 : 0; }
    si_prof_begin(ctx, name, 2.0 * rows * (double)CIN * N * UP_TAPS, 2.0 * rows * ((double)CIN + N) + 2.0 * UP_TAPS * N * CIN, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * 4 * WN), lds, st, pk);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller runs the tap-GEMM).
int si_launch_upsample_stream(si_ctx* ctx, const UpsampleParams& p, hipStream_t st) {
    if (p.taps != UP_TAPS || p.N != p.Cin || !p.x16 || !p.out16 || !p.w || !p.bias || p.B <= 0 || p.M <= 0) return 1;
    if (p.lens_m && (!p.lens_lin || !p.lens_m_host)) return si_fail(ctx, SI_EINVAL, "upsample: ragged batches need lens_lin, lens_m and lens_m_host");
    if (p.ooff % 8 || p.o_clip_elems % 8 || (long)p.Lin * p.Cin * 2 >= (1L << 31) || p.o_clip_elems * 2 >= (1L << 31) ||
        ((long)p.M + 256) * p.N * 2 >= (1L << 31)) return 1;
    if (p.Cin == 128) return upsample_launch<128, 128>(ctx, p, st);
    if (p.Cin == 64) return upsample_launch<64, 64>(ctx, p, st);
    return 1;
}
