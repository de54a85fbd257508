// tapgemm.hip -- the contraction kernel of the path (gfx950, wave64, MFMA).
//
// Computes, on channels-last fp32 activations,
//     out[seg][m][n] = epi( sum_tap sum_ci pro(x[seg][m*stride + tap*dil - pad][g*Cin + ci]) * W[g][tap][n][ci] )
// which covers Conv1d (stride / dilation / groups / zero padding), Linear (ntaps = 1) and ConvTranspose1d
// (split into its `stride` output phases: 2 taps with dil = -1, N = stride*Cout, see api.hip).
// Replaces the torch.nn calls of SURVEY.md 8(a) rows A2-A9, B1-B3.
//
// Structure: one 256-thread workgroup (4 waves) owns a BM x BN output tile.  For every K chunk of BK input
// channels the halo'd activation tile ((BM-1)*stride + (ntaps-1)*|dil| + 1 rows) is staged ONCE into LDS and
// reused by all taps -- a tap is just a row offset into that tile -- while the per-tap BK x BN weight slab
// (L2-resident, shared by every workgroup) is re-staged per tap.  The prologue activation (leaky-relu) and the
// fp32 -> bf16 / bf16 hi+lo conversion happen while staging, the epilogue (bias, GELU, residual, scale,
// accumulate) on the accumulators.
//
// MFMA use (cdna_hip_programming.md section 3):
//   F32    v_mfma_f32_32x32x2_f32 : lane l supplies A[l&31][k=l>>5], B[k=l>>5][l&31].  The k index is a
//          dummy, so lane-half h is given the K range [h*BK/2, (h+1)*BK/2) of the chunk: consecutive MFMA
//          steps then read consecutive floats and one ds_read_b128 feeds four steps.
//   BF16   v_mfma_f32_32x32x16_bf16: lane supplies 8 consecutive k (16 B) of row l&31, k block l>>5.
//   BF16X3 same instruction three times (hi*hi + lo*hi + hi*lo) for ~fp32 accuracy at 3/16 of the fp32 cost.
// LDS rows are padded by 16 B so the 16 rows a ds_read_b128 lane group touches fall on distinct 4-bank slots.
#include <cstdio>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int MATH> struct LdsElem { typedef float type; static constexpr int PAD = 4; };
template <> struct LdsElem<SI_MATH_BF16> { typedef unsigned short type; static constexpr int PAD = 8; };
template <> struct LdsElem<SI_MATH_BF16X3> { typedef unsigned short type; static constexpr int PAD = 8; };

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK>
__global__ __launch_bounds__(256) void tapgemm_kernel(const TapGemmParams p) {
    static_assert(WARPS_M * WARPS_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile is a multiple of 32x32");
    typedef typename LdsElem<MATH>::type elem_t;
    constexpr int LD = BK + LdsElem<MATH>::PAD;          // LDS row stride in elements (row = BK*sizeof + 16 B)
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;

    const int mtiles = (p.M + BM - 1) / BM;
    const int seg = blockIdx.x / mtiles;
    const int m0 = (blockIdx.x % mtiles) * BM;
    const int n0 = blockIdx.y * BN;
    const int g = blockIdx.z;

    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int dil_lo = p.dil < 0 ? (p.ntaps - 1) * p.dil : 0;
    const int base_in = m0 * p.stride - p.pad + dil_lo;          // input row held in LDS row 0
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;

    elem_t* As = reinterpret_cast<elem_t*>(smem);                 // [PLANES][rowsA][LD]
    elem_t* Bs = As + (size_t)PLANES * rowsA * LD;                // [PLANES][BN][LD]

    const float* xs = p.x + (long)seg * p.x_seg_stride + (long)g * p.Cin;
    const size_t wplane = (size_t)p.ntaps * p.Npad * p.Cin;      // elements per group
    const float slope = p.pro_slope;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int c0 = 0; c0 < p.Cin; c0 += BK) {
        __syncthreads();                                          // everyone is done reading As / Bs
        // ---- stage the halo'd activation tile (all taps read it) ----
        constexpr int V4 = BK / 4;
        for (int idx = tid; idx < rowsA * V4; idx += 256) {
            const int r = idx / V4, j = idx - r * V4;
            const int grow = base_in + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (grow >= 0 && grow < p.Lin) v = *reinterpret_cast<const f32x4*>(xs + (long)grow * p.ldx + c0 + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
            if constexpr (MATH == SI_MATH_F32) {
                *reinterpret_cast<f32x4*>(As + r * LD + 4 * j) = v;
            } else {
                u16x4 hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) hi[e] = f2bf_rne(v[e]);
                *reinterpret_cast<u16x4*>(As + r * LD + 4 * j) = hi;
                if constexpr (MATH == SI_MATH_BF16X3) {
                    u16x4 lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) lo[e] = f2bf_rne(v[e] - bf2f(hi[e]));
                    *reinterpret_cast<u16x4*>(As + (size_t)rowsA * LD + r * LD + 4 * j) = lo;
                }
            }
        }
        for (int tap = 0; tap < p.ntaps; ++tap) {
            if (tap > 0) __syncthreads();                         // previous tap's MFMAs are done with Bs
            // ---- stage this tap's weight slab W[g][tap][n0 .. n0+BN)[c0 .. c0+BK) ----
            if constexpr (MATH == SI_MATH_F32) {
                const float* wg = reinterpret_cast<const float*>(p.w) + (size_t)g * wplane +
                                  ((size_t)tap * p.Npad + n0) * p.Cin + c0;
                for (int idx = tid; idx < BN * V4; idx += 256) {
                    const int r = idx / V4, j = idx - r * V4;
                    *reinterpret_cast<f32x4*>(Bs + r * LD + 4 * j) =
                        *reinterpret_cast<const f32x4*>(wg + (size_t)r * p.Cin + 4 * j);
                }
            } else {
                constexpr int V8 = BK / 8;
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const unsigned short* wg = reinterpret_cast<const unsigned short*>(pl == 0 ? p.w : p.w_lo) +
                                               (size_t)g * wplane + ((size_t)tap * p.Npad + n0) * p.Cin + c0;
                    for (int idx = tid; idx < BN * V8; idx += 256) {
                        const int r = idx / V8, j = idx - r * V8;
                        *reinterpret_cast<u16x8*>(Bs + (size_t)pl * BN * LD + r * LD + 8 * j) =
                            *reinterpret_cast<const u16x8*>(wg + (size_t)r * p.Cin + 8 * j);
                    }
                }
            }
            __syncthreads();
            // ---- MFMA over this (chunk, tap) ----
            const int toff = tap * p.dil - dil_lo;               // LDS row offset of this tap (>= 0)
            if constexpr (MATH == SI_MATH_F32) {
                const float* ap[TM];
                const float* bp[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) ap[i] = As + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * (BK / 2);
#pragma unroll
                for (int j = 0; j < TN; ++j) bp[j] = Bs + (wn0 + j * 32 + l31) * LD + half * (BK / 2);
#pragma unroll
                for (int s4 = 0; s4 < BK / 8; ++s4) {
                    f32x4 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap[i] + 4 * s4);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bp[j] + 4 * s4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
                }
            } else {
                const unsigned short* ap[TM];
                const unsigned short* bp[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) ap[i] = As + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * 8;
#pragma unroll
                for (int j = 0; j < TN; ++j) bp[j] = Bs + (wn0 + j * 32 + l31) * LD + half * 8;
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    bf16x8 ah[TM], bh[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const bf16x8*>(ap[i] + 16 * ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const bf16x8*>(bp[j] + 16 * ks);
                    if constexpr (MATH == SI_MATH_BF16X3) {
                        bf16x8 al[TM], bl[TN];
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            al[i] = *reinterpret_cast<const bf16x8*>(ap[i] + (size_t)rowsA * LD + 16 * ks);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bl[j] = *reinterpret_cast<const bf16x8*>(bp[j] + (size_t)BN * LD + 16 * ks);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                // small terms first so they are not swamped by the running sum
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const long obase = (long)seg * p.o_seg_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[g * p.N + n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.M) continue;
                const long flat = (long)m * p.ldo + (long)g * p.N + n + p.ooff;
                if (flat < 0 || flat >= p.olimit) continue;
                float v = acc[i][j][r] + bv;
                if (p.act == SI_ACT_GELU) v = gelu_erf(v);
                if (p.res) v += p.res[obase + flat];
                v *= p.alpha;
                if (p.accumulate) v += p.out[obase + flat];
                p.out[obase + flat] = v;
            }
        }
    }
}

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK>
static int launch_cfg(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    typedef typename LdsElem<MATH>::type elem_t;
    constexpr int LD = BK + LdsElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const size_t lds = (size_t)PLANES * (rowsA + BN) * LD * sizeof(elem_t);
    if (lds > 160 * 1024) return si_fail(ctx, SI_EINVAL, "tapgemm: LDS tile of %zu bytes exceeds 160 KiB", lds);
    auto kern = tapgemm_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK>;
    if (lds > 64 * 1024) {
        SI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int mtiles = (p.M + BM - 1) / BM;
    dim3 grid((unsigned)(p.nseg * mtiles), (unsigned)((p.N + BN - 1) / BN), (unsigned)p.groups);
    static const char* const math_names[] = {"f32", "bf16", "bf16x3"};
    char name[48];
    snprintf(name, sizeof(name), "tapgemm_%s_%dx%d", math_names[MATH], BM, BN);
    const double macs = p.algo_macs > 0 ? p.algo_macs : (double)p.nseg * p.M * p.N * p.groups * (double)p.Cin * p.ntaps;
    double bytes = 4.0 * p.nseg * ((double)p.Lin * p.Cin * p.groups + (double)p.M * p.N * p.groups * (1 + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0))) +
                   (double)p.groups * p.ntaps * p.N * p.Cin * (MATH == SI_MATH_F32 ? 4 : (MATH == SI_MATH_BF16 ? 2 : 4));
    si_prof_begin(ctx, name, 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

template <int MATH, int BK>
static int launch_math(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    const int bn = si_pick_bn(p.N);
    if (bn == 128) return launch_cfg<MATH, 128, 128, 2, 2, BK>(ctx, p, st);
    if (bn == 64) return launch_cfg<MATH, 256, 64, 4, 1, BK>(ctx, p, st);
    return launch_cfg<MATH, 256, 32, 4, 1, BK>(ctx, p, st);
}

int si_launch_tapgemm(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st) {
    if (p.Cin % 16 != 0 || p.ldx % 4 != 0)
        return si_fail(ctx, SI_EINVAL, "tapgemm: Cin=%d must be a multiple of 16 and ldx=%d of 4", p.Cin, p.ldx);
    if (p.Npad % si_pick_bn(p.N) != 0 || p.Npad < p.N)
        return si_fail(ctx, SI_EINVAL, "tapgemm: Npad=%d does not match N=%d", p.Npad, p.N);
    if (p.M <= 0 || p.nseg <= 0) return SI_OK;
    const bool k32 = (p.Cin % 32 == 0);
    switch (math) {
        case SI_MATH_F32: return k32 ? launch_math<SI_MATH_F32, 32>(ctx, p, st) : launch_math<SI_MATH_F32, 16>(ctx, p, st);
        case SI_MATH_BF16: return k32 ? launch_math<SI_MATH_BF16, 32>(ctx, p, st) : launch_math<SI_MATH_BF16, 16>(ctx, p, st);
        case SI_MATH_BF16X3: return k32 ? launch_math<SI_MATH_BF16X3, 32>(ctx, p, st) : launch_math<SI_MATH_BF16X3, 16>(ctx, p, st);
    }
    return si_fail(ctx, SI_EINVAL, "tapgemm: unknown math mode %d", math);
}
