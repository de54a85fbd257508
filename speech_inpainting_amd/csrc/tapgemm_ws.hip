// tapgemm_ws.hip -- wave-specialised form of the tap-GEMM contraction kernel (gfx950, wave64, MFMA).
//
// Same contraction, operands, LDS images and epilogue as tapgemm.hip; what changes is WHO moves data.
// An ablation of tapgemm.hip (MFMA section compiled out) showed that global->LDS staging + epilogue of the
// 128x128 family cost 10.8 ms per bench step in every arithmetic mode, and that this time ADDS to the MFMA time
// instead of hiding under it: each wave alternates between issuing loads / converting / writing LDS and issuing
// MFMAs, loads have at most one (short, in bf16 modes) iteration to land, and a wave waiting on memory issues no
// MFMAs.  Here a 512-thread workgroup is split by wave role:
//   waves 0-3  consumers : ds_read fragments + MFMA only, then the epilogue
//   waves 4-7  producers : global loads -> (leaky-relu, bf16 / hi+lo split) -> LDS writes, nothing else
// A workgroup's waves are dealt to the SIMDs cyclically, so every SIMD hosts one consumer and one producer: the
// producer's VMEM / VALU / LDS-write instructions co-issue beside the consumer's MFMAs instead of interrupting
// them.  Producers run ahead: two weight slabs are in flight (iteration it+2 is issued while it+1 lands), and
// the next activation chunk is issued at the first tap of the current chunk, so every load has at least a full
// iteration -- for convolutions ntaps-1 iterations -- to land.  Both LDS images are double-buffered; there is
// one workgroup barrier per (chunk, tap) iteration and no other synchronisation.  All global loads are ordinary
// register loads, so the compiler's counted s_waitcnt vmcnt(N) keeps exactly the younger batches in flight.
#include <cstdio>
#include <cstdlib>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ws_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int MATH> struct WsElem { typedef float type; static constexpr int PAD = 4; };
template <> struct WsElem<SI_MATH_BF16> { typedef unsigned short type; static constexpr int PAD = 8; };
template <> struct WsElem<SI_MATH_BF16X3> { typedef unsigned short type; static constexpr int PAD = 8; };

#define WS_MAXA 12      // float4 of one activation chunk a producer thread holds in flight (rowsA * BK/4 <= 3072)

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK>
__global__ __launch_bounds__(512) void tapgemm_ws_kernel(const TapGemmParams p) {
    static_assert(WARPS_M * WARPS_N == 4, "4 consumer waves per workgroup");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    typedef typename WsElem<MATH>::type elem_t;
    constexpr int LD = BK + WsElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    constexpr int V4 = BK / 4;
    constexpr int VB = (MATH == SI_MATH_F32) ? BK / 4 : BK / 8;
    constexpr int MAXB = (BN * VB + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int stid = tid & 255;                                   // thread index inside its role group

    const int mtiles = (p.M + BM - 1) / BM;
    const int ntn = (p.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn;
    const int seg = mt / mtiles;
    const int m0 = (mt % mtiles) * BM;
    const int n0 = (blockIdx.x % ntn) * BN;
    const int g = blockIdx.y;

    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int dil_lo = p.dil < 0 ? (p.ntaps - 1) * p.dil : 0;
    const int base_in = m0 * p.stride - p.pad + dil_lo;
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const int nA = rowsA * V4;
    const int ntaps = p.ntaps;
    const int nchunks = p.Cin / BK;
    const int total = nchunks * ntaps;

    const size_t a_tile = (size_t)PLANES * rowsA * LD;
    constexpr size_t b_tile = (size_t)PLANES * BN * LD;
    elem_t* As = reinterpret_cast<elem_t*>(smem);                 // [2][PLANES][rowsA][LD]
    elem_t* Bs = As + 2 * a_tile;                                 // [2][PLANES][BN][LD]

    if (producer) {
        // ======================================================================================= producers
        const float* xs = p.x + (long)seg * p.x_seg_stride + (long)g * p.Cin;
        const size_t wplane = (size_t)ntaps * p.Npad * p.Cin;
        const float slope = p.pro_slope;
        f32x4 ra[WS_MAXA];
        f32x4 rb0[PLANES][MAXB], rb1[PLANES][MAXB];

        auto issueA = [&](int c0) {
#pragma unroll
            for (int i = 0; i < WS_MAXA; ++i) {
                const int idx = stid + i * 256;
                ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < nA) {
                    const int r = idx / V4, j = idx - r * V4;
                    const int grow = base_in + r;
                    if (grow >= 0 && grow < p.Lin) ra[i] = *reinterpret_cast<const f32x4*>(xs + (long)grow * p.ldx + c0 + 4 * j);
                }
            }
        };
        auto storeA = [&](elem_t* dst) {
#pragma unroll
            for (int i = 0; i < WS_MAXA; ++i) {
                const int idx = stid + i * 256;
                if (idx < nA) {
                    const int r = idx / V4, j = idx - r * V4;
                    f32x4 v = ra[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                    if constexpr (MATH == SI_MATH_F32) {
                        *reinterpret_cast<f32x4*>(dst + r * LD + 4 * j) = v;
                    } else {
                        const bf16x4 hi = __builtin_convertvector(v, bf16x4);          // v_cvt_pk_bf16_f32 (RNE)
                        *reinterpret_cast<bf16x4*>(dst + r * LD + 4 * j) = hi;
                        if constexpr (MATH == SI_MATH_BF16X3) {
                            const f32x4 rem = v - __builtin_convertvector(hi, f32x4);
                            *reinterpret_cast<bf16x4*>(dst + (size_t)rowsA * LD + r * LD + 4 * j) = __builtin_convertvector(rem, bf16x4);
                        }
                    }
                }
            }
        };
        auto issueB = [&](f32x4 (&rb)[PLANES][MAXB], int c0, int tap) {
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                const char* wbase = reinterpret_cast<const char*>(pl == 0 ? p.w : p.w_lo) +
                                    sizeof(elem_t) * ((size_t)g * wplane + ((size_t)tap * p.Npad + n0) * p.Cin + c0);
#pragma unroll
                for (int i = 0; i < MAXB; ++i) {
                    const int idx = stid + i * 256;
                    if (BN * VB % 256 == 0 || idx < BN * VB) {
                        const int r = idx / VB, j = idx - r * VB;
                        rb[pl][i] = *reinterpret_cast<const f32x4*>(wbase + sizeof(elem_t) * (size_t)r * p.Cin + 16 * j);
                    }
                }
            }
        };
        auto storeB = [&](const f32x4 (&rb)[PLANES][MAXB], elem_t* dst) {
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int i = 0; i < MAXB; ++i) {
                    const int idx = stid + i * 256;
                    if (BN * VB % 256 == 0 || idx < BN * VB) {
                        const int r = idx / VB, j = idx - r * VB;
                        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst + (size_t)pl * BN * LD + r * LD) + 16 * j) = rb[pl][i];
                    }
                }
        };

        // descriptors of iterations it (c0,t0), it+1 (c1,t1), it+2 (c2,t2)
        int c0 = 0, t0 = 0, c1 = 0, t1 = 0, c2 = 0, t2 = 0;
        auto adv = [&](int& c, int& t) { if (++t == ntaps) { t = 0; ++c; } };
        adv(c1, t1);
        adv(c2, t2); adv(c2, t2);

        issueA(0);
        issueB(rb0, 0, 0);
        if (total > 1) issueB(rb1, c1 * BK, t1);
        storeA(As);
        storeB(rb0, Bs);
        __syncthreads();                                           // barrier 0: iteration 0 is staged

        // one step: `rissue` receives B(it+2), `rland` holds B(it+1) and is written to LDS
        auto step = [&](f32x4 (&rissue)[PLANES][MAXB], const f32x4 (&rland)[PLANES][MAXB], int it) {
            if (it + 2 < total) issueB(rissue, c2 * BK, t2);
            if (t0 == 0 && c0 + 1 < nchunks) issueA((c0 + 1) * BK);          // next chunk: ntaps-1 iterations to land
            if (it + 1 < total) {
                storeB(rland, Bs + (size_t)((it + 1) & 1) * b_tile);
                if (t1 == 0) storeA(As + (size_t)(c1 & 1) * a_tile);          // next iteration opens a new chunk
            }
            __syncthreads();
            c0 = c1; t0 = t1; c1 = c2; t1 = t2;
            adv(c2, t2);
        };
        for (int it = 0; it < total; it += 2) {
            step(rb0, rb1, it);                                    // even: rb0 was landed last step (or in the prologue)
            if (it + 1 < total) step(rb1, rb0, it + 1);
        }
        return;
    }

    // =========================================================================================== consumers
    const int lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __syncthreads();                                               // barrier 0
    int chunk = 0, tap = 0;
    for (int it = 0; it < total; ++it) {
        const elem_t* Ac = As + (size_t)(chunk & 1) * a_tile;
        const elem_t* Bc = Bs + (size_t)(it & 1) * b_tile;
        const int toff = tap * p.dil - dil_lo;
        if constexpr (MATH == SI_MATH_F32) {
            const float* ap[TM];
            const float* bp[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ap[i] = Ac + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * (BK / 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) bp[j] = Bc + (wn0 + j * 32 + l31) * LD + half * (BK / 2);
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
            for (int i = 0; i < TM; ++i) ap[i] = Ac + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * 8;
#pragma unroll
            for (int j = 0; j < TN; ++j) bp[j] = Bc + (wn0 + j * 32 + l31) * LD + half * 8;
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
                    for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const bf16x8*>(ap[i] + (size_t)rowsA * LD + 16 * ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(bp[j] + (size_t)BN * LD + 16 * ks);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
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
        __syncthreads();
        if (++tap == ntaps) { tap = 0; ++chunk; }
    }

    // ---- epilogue (as tapgemm.hip): batched, branch-free residual / accumulate reads, 32-bit in-segment offsets ----
    float* const outp = p.out + (long)seg * p.o_seg_stride;
    const float* const resp = p.res ? p.res + (long)seg * p.o_seg_stride : nullptr;
    const bool has_res = p.res != nullptr;
    const bool acc_out = p.accumulate != 0;
    const bool gelu = p.act == SI_ACT_GELU;
    const int olim = (int)p.olimit;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            const bool nok = n < p.N;
            const float bv = (p.bias && nok) ? p.bias[g * p.N + n] : 0.f;
            const int col = g * p.N + n + (int)p.ooff;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                int fl[8];
                unsigned okm = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = hb * 8 + q;
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const long flat = (long)m * p.ldo + col;
                    const bool ok = nok && m < p.M && flat >= 0 && flat < olim;
                    okm |= (ok ? 1u : 0u) << q;
                    fl[q] = ok ? (int)flat : 0;
                }
                float rv[8], ov[8];
                if (has_res) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) rv[q] = resp[fl[q]];
                }
                if (acc_out) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) ov[q] = outp[fl[q]];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = acc[i][j][hb * 8 + q] + bv;
                    if (gelu) v = ws_gelu_erf(v);
                    if (has_res) v += rv[q];
                    v *= p.alpha;
                    if (acc_out) v += ov[q];
                    if ((okm >> q) & 1u) outp[fl[q]] = v;
                }
            }
        }
    }
}

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK>
static int ws_launch_cfg(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    typedef typename WsElem<MATH>::type elem_t;
    constexpr int LD = BK + WsElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const size_t lds = (size_t)PLANES * (2 * (size_t)rowsA + 2 * BN) * LD * sizeof(elem_t);
    if (rowsA * (BK / 4) > WS_MAXA * 256 || lds > 160 * 1024) return 1;          // not applicable: caller falls back
    auto kern = tapgemm_ws_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK>;
    static size_t lds_set = 0;                                                     // per instantiation
    if (lds > 64 * 1024 && lds > lds_set) {
        SI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    const int mtiles = (p.M + BM - 1) / BM;
    dim3 grid((unsigned)(p.nseg * mtiles * ((p.N + BN - 1) / BN)), (unsigned)p.groups);
    static const char* const math_names[] = {"f32", "bf16", "bf16x3"};
    char name[48];
    snprintf(name, sizeof(name), "tapgemm_ws_%s_%dx%d", math_names[MATH], BM, BN);
    const double macs = p.algo_macs > 0 ? p.algo_macs : (double)p.nseg * p.M * p.N * p.groups * (double)p.Cin * p.ntaps;
    double bytes = 4.0 * p.nseg * ((double)p.Lin * p.Cin * p.groups + (double)p.M * p.N * p.groups * (1 + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0))) +
                   (double)p.groups * p.ntaps * p.N * p.Cin * (MATH == SI_MATH_F32 ? 4 : (MATH == SI_MATH_BF16 ? 2 : 4));
    si_prof_begin(ctx, name, 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

template <int MATH>
static int ws_launch_math(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    const int bn = si_pick_bn(p.N);
    if (bn == 128) return ws_launch_cfg<MATH, 128, 128, 2, 2, 32>(ctx, p, st);
    if (p.M <= 128) return 1;
    if (bn == 64) return ws_launch_cfg<MATH, 256, 64, 4, 1, 32>(ctx, p, st);
    return ws_launch_cfg<MATH, 256, 32, 4, 1, 32>(ctx, p, st);
}

// Returns SI_OK when launched, a negative SI_E* on error, and 1 when this shape is not covered (caller falls back
// to the unified kernel of tapgemm.hip): Cin not a multiple of 32, or a tile that does not fit.
int si_launch_tapgemm_ws(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st) {
    if (p.Cin % 32 != 0) return 1;
    switch (math) {
        case SI_MATH_F32: return ws_launch_math<SI_MATH_F32>(ctx, p, st);
        case SI_MATH_BF16: return ws_launch_math<SI_MATH_BF16>(ctx, p, st);
        case SI_MATH_BF16X3: return ws_launch_math<SI_MATH_BF16X3>(ctx, p, st);
    }
    return 1;
}
