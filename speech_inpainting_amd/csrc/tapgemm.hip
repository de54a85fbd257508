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
// Pipeline: the loop runs over (chunk, tap) iterations.  While iteration `it` issues its MFMAs out of LDS, the
// global loads of iteration it+1's weight slab (and, at a chunk boundary, of the next activation tile) are
// already in flight into registers; they are written to LDS after the MFMAs, into the OTHER weight buffer, so
// there is one workgroup barrier per iteration.  The activation tile is single-buffered for convolutions (its
// halo makes it large; a chunk boundary costs one extra barrier every `ntaps` iterations) and double-buffered
// for ntaps == 1 (Linear layers, where every iteration is a chunk boundary).
//
// MFMA use (cdna_hip_programming.md section 3):
//   F32    v_mfma_f32_32x32x2_f32 : lane l supplies A[l&31][k=l>>5], B[k=l>>5][l&31].  The k index is a
//          dummy, so lane-half h is given the K range [h*BK/2, (h+1)*BK/2) of the chunk: consecutive MFMA
//          steps then read consecutive floats and one ds_read_b128 feeds four steps.
//   BF16   v_mfma_f32_32x32x16_bf16: lane supplies 8 consecutive k (16 B) of row l&31, k block l>>5.
//   BF16X3 same instruction three times (lo*hi + hi*lo + hi*hi) for ~fp32 accuracy at 3/16 of the fp32 cost.
// LDS rows are padded by 16 B so the 16 rows a ds_read_b128 lane group touches fall on distinct 4-bank slots.
#include <cstdio>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
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

#define TG_MAXA 12      // float4 of the activation tile a thread can hold in flight (rowsA * BK/4 <= 3072)

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK>
__global__ __launch_bounds__(256) void tapgemm_kernel(const TapGemmParams p) {
    static_assert(WARPS_M * WARPS_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile is a multiple of 32x32");
    typedef typename LdsElem<MATH>::type elem_t;
    constexpr int LD = BK + LdsElem<MATH>::PAD;          // LDS row stride in elements (row = BK*sizeof + 16 B)
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    constexpr int V4 = BK / 4;                           // float4 per activation row
    constexpr int VB = (MATH == SI_MATH_F32) ? BK / 4 : BK / 8;     // 16-byte vectors per weight row (per plane)
    constexpr int MAXB = (BN * VB + 255) / 256;          // 16-byte vectors of a weight slab per thread (per plane)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;

    const int mtiles = (p.M + BM - 1) / BM;
    // N-tiles of one M-tile are adjacent in dispatch order: they read the same activation rows, so all but the
    // first find them in L2 / Infinity Cache instead of HBM.
    const int ntn = (p.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn;
    const int seg = mt / mtiles;
    const int m0 = (mt % mtiles) * BM;
    const int n0 = (blockIdx.x % ntn) * BN;
    const int g = blockIdx.y;

    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int dil_lo = p.dil < 0 ? (p.ntaps - 1) * p.dil : 0;
    const int base_in = m0 * p.stride - p.pad + dil_lo;          // input row held in LDS row 0
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const int nA = rowsA * V4;                                    // float4 in one activation tile
    const int abufs = p.ntaps == 1 ? 2 : 1;

    const size_t a_tile = (size_t)PLANES * rowsA * LD;            // elements per activation buffer
    constexpr size_t b_tile = (size_t)PLANES * BN * LD;           // elements per weight buffer
    elem_t* As = reinterpret_cast<elem_t*>(smem);                 // [abufs][PLANES][rowsA][LD]
    elem_t* Bs = As + (size_t)abufs * a_tile;                     // [2][PLANES][BN][LD]

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

    f32x4 ra[TG_MAXA];                    // activation tile in flight
    f32x4 rb[PLANES][MAXB];               // weight slab in flight (16 bytes each, whatever the element type)

    auto issueA = [&](int c0) {
#pragma unroll
        for (int i = 0; i < TG_MAXA; ++i) {
            const int idx = tid + i * 256;
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
        for (int i = 0; i < TG_MAXA; ++i) {
            const int idx = tid + i * 256;
            if (idx < nA) {
                const int r = idx / V4, j = idx - r * V4;
                f32x4 v = ra[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                if constexpr (MATH == SI_MATH_F32) {
                    *reinterpret_cast<f32x4*>(dst + r * LD + 4 * j) = v;
                } else {
                    // vector casts lower to v_cvt_pk_bf16_f32 (round-to-nearest-even, 2 elements per instruction);
                    // a bit-twiddled RNE costs ~15 VALU ops per element, which in bf16x3 mode rivalled the MFMA
                    // time of a whole chunk
                    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                    *reinterpret_cast<bf16x4*>(dst + r * LD + 4 * j) = hi;
                    if constexpr (MATH == SI_MATH_BF16X3) {
                        const f32x4 rem = v - __builtin_convertvector(hi, f32x4);
                        *reinterpret_cast<bf16x4*>(dst + (size_t)rowsA * LD + r * LD + 4 * j) = __builtin_convertvector(rem, bf16x4);
                    }
                }
            }
        }
    };
    auto issueB = [&](int c0, int tap) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            const char* wbase = reinterpret_cast<const char*>(pl == 0 ? p.w : p.w_lo) +
                                sizeof(elem_t) * ((size_t)g * wplane + ((size_t)tap * p.Npad + n0) * p.Cin + c0);
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int idx = tid + i * 256;
                if (BN * VB % 256 == 0 || idx < BN * VB) {
                    const int r = idx / VB, j = idx - r * VB;
                    rb[pl][i] = *reinterpret_cast<const f32x4*>(wbase + sizeof(elem_t) * (size_t)r * p.Cin + 16 * j);
                }
            }
        }
    };
    auto storeB = [&](elem_t* dst) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int idx = tid + i * 256;
                if (BN * VB % 256 == 0 || idx < BN * VB) {
                    const int r = idx / VB, j = idx - r * VB;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst + (size_t)pl * BN * LD + r * LD) + 16 * j) = rb[pl][i];
                }
            }
    };

    const int nchunks = p.Cin / BK;
    const int total = nchunks * p.ntaps;

    // (Measured and rejected: delaying the second resident workgroup of each CU by 1/2 or 1 tile to break the
    // lock-step of co-resident workgroups cost 3-5 % on every fp32 shape.)
    issueA(0);
    issueB(0, 0);
    storeA(As);
    storeB(Bs);
    __syncthreads();

    int chunk = 0, tap = 0;
    for (int it = 0; it < total; ++it) {
        const bool has_next = it + 1 < total;
        const bool new_chunk = has_next && (tap == p.ntaps - 1);
        const int ntap = new_chunk ? 0 : tap + 1;
        const int nchunk = new_chunk ? chunk + 1 : chunk;
        if (has_next) issueB(nchunk * BK, ntap);
        if (new_chunk) issueA(nchunk * BK);

        // ---- MFMA over this (chunk, tap) out of LDS ----
        const elem_t* Ac = As + (size_t)(abufs == 2 ? (chunk & 1) : 0) * a_tile;
        const elem_t* Bc = Bs + (size_t)(it & 1) * b_tile;
        const int toff = tap * p.dil - dil_lo;                   // LDS row offset of this tap (>= 0)
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

        // ---- land the prefetched tiles ----
        if (has_next) {
            if (new_chunk && abufs == 1) __syncthreads();         // every wave is done reading the activation tile
            storeB(Bs + (size_t)((it + 1) & 1) * b_tile);
            if (new_chunk) storeA(As + (size_t)(abufs == 2 ? (nchunk & 1) : 0) * a_tile);
            __syncthreads();
        }
        tap = ntap;
        chunk = nchunk;
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    // Residual / accumulate reads are issued as one unconditional batch per 32x32 tile (out-of-range elements
    // read the segment's element 0 and are dropped at the store): a per-element predicated read serialises 16
    // dependent global round trips per tile and doubled the time of every residual conv.
    // Offsets inside a segment are 32-bit (the launcher checks olimit < 2^31) off a workgroup-uniform base.
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
            for (int hb = 0; hb < 2; ++hb) {                       // two batches of 8 keep the register footprint small
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
                    if (gelu) v = gelu_erf(v);
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
static int launch_cfg(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    typedef typename LdsElem<MATH>::type elem_t;
    constexpr int LD = BK + LdsElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    if (rowsA * (BK / 4) > TG_MAXA * 256)
        return si_fail(ctx, SI_EINVAL, "tapgemm: activation tile of %d rows exceeds the prefetch registers (stride %d, taps %d, dil %d)",
                       rowsA, p.stride, p.ntaps, p.dil);
    const int abufs = p.ntaps == 1 ? 2 : 1;
    const size_t lds = (size_t)PLANES * ((size_t)abufs * rowsA + 2 * BN) * LD * sizeof(elem_t);
    if (lds > 160 * 1024) return si_fail(ctx, SI_EINVAL, "tapgemm: LDS tile of %zu bytes exceeds 160 KiB", lds);
    auto kern = tapgemm_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK>;
    if (lds > 64 * 1024) {
        SI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int mtiles = (p.M + BM - 1) / BM;
    dim3 grid((unsigned)(p.nseg * mtiles * ((p.N + BN - 1) / BN)), (unsigned)p.groups);
    static const char* const math_names[] = {"f32", "bf16", "bf16x3"};
    char name[48];
    snprintf(name, sizeof(name), "tapgemm_%s_%dx%d%s", math_names[MATH], BM, BN, BK == 64 ? "k64" : "");
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
    // narrow N: 256-row tiles unless their halo'd activation tile would not fit the prefetch registers
    // (strided convs), or the segment is so short that a 256-row tile would be mostly padding
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const bool tall = ((255 * p.stride + (p.ntaps - 1) * adil + 1) * (BK / 4) <= TG_MAXA * 256) && p.M > 128;
    if (bn == 64) return tall ? launch_cfg<MATH, 256, 64, 4, 1, BK>(ctx, p, st) : launch_cfg<MATH, 128, 64, 2, 2, BK>(ctx, p, st);
    return tall ? launch_cfg<MATH, 256, 32, 4, 1, BK>(ctx, p, st) : launch_cfg<MATH, 128, 32, 4, 1, BK>(ctx, p, st);
}

int si_launch_tapgemm(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st) {
    if (p.Cin % 16 != 0 || p.ldx % 4 != 0)
        return si_fail(ctx, SI_EINVAL, "tapgemm: Cin=%d must be a multiple of 16 and ldx=%d of 4", p.Cin, p.ldx);
    if (p.Npad % si_pick_bn(p.N) != 0 || p.Npad < p.N)
        return si_fail(ctx, SI_EINVAL, "tapgemm: Npad=%d does not match N=%d", p.Npad, p.N);
    if (p.M <= 0 || p.nseg <= 0) return SI_OK;
    if (p.olimit >= (1L << 31) || (long)p.M * p.ldo + p.ooff >= (1L << 31) || (long)p.Lin * p.ldx >= (1L << 31))
        return si_fail(ctx, SI_EINVAL, "tapgemm: a segment of %ld floats exceeds the 32-bit in-segment offsets", p.olimit);
    const bool k32 = (p.Cin % 32 == 0);
    // bf16 MFMAs retire a 32-deep K chunk in a quarter of the fp32 time, so the barrier + staging cost per chunk
    // dominates; a 64-deep chunk halves it where the (wider) activation tile still fits the prefetch registers.
    static const int bk64_mode = getenv("SI_TG_BK64") ? atoi(getenv("SI_TG_BK64")) : 0;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int bm = si_pick_bn(p.N) == 128 ? 128 : 256;
    const bool k64 = bk64_mode && math != SI_MATH_F32 && p.Cin % 64 == 0 &&
                     ((bm - 1) * p.stride + (p.ntaps - 1) * adil + 1) * 16 <= TG_MAXA * 256;
    switch (math) {
        case SI_MATH_F32: return k32 ? launch_math<SI_MATH_F32, 32>(ctx, p, st) : launch_math<SI_MATH_F32, 16>(ctx, p, st);
        case SI_MATH_BF16:
            if (k64) return launch_math<SI_MATH_BF16, 64>(ctx, p, st);
            return k32 ? launch_math<SI_MATH_BF16, 32>(ctx, p, st) : launch_math<SI_MATH_BF16, 16>(ctx, p, st);
        case SI_MATH_BF16X3:
            if (k64) return launch_math<SI_MATH_BF16X3, 64>(ctx, p, st);
            return k32 ? launch_math<SI_MATH_BF16X3, 32>(ctx, p, st) : launch_math<SI_MATH_BF16X3, 16>(ctx, p, st);
    }
    return si_fail(ctx, SI_EINVAL, "tapgemm: unknown math mode %d", math);
}
