// tapgemm_pp.hip -- "ping-pong" form of the tap-GEMM convolution kernel (gfx950, wave64, MFMA).
//
// Same contraction, LDS images, MFMA use and epilogue arithmetic as tapgemm.hip; what changes is WHO does what WHEN.
// PMC counters on tapgemm.hip's bf16x3 256x128 8-wave tile show the matrix pipe busy only 43 % of the time
// (fp32: 79 %): its eight waves run the same program between the same barriers, so the two waves that share a SIMD
// reach their MFMA section together (and queue on the one matrix pipe) and their staging section together (and leave
// the pipe idle).  MI355X_MICROARCH.md ("Two waves per SIMD") describes the remedy used by tuned 8-wave attention
// loops: give the two waves of a SIMD COMPLEMENTARY segments.
//
// Here the workgroup's waves form two groups (waves 0-3 = rows 0..127 of the 256x128 tile, waves 4-7 = rows
// 128..255; a workgroup's waves 0-3 and 4-7 each cover the four SIMDs).  Every (chunk, tap) iteration has two phases
// separated by a barrier:
//     phase 2*it     : group 0 issues its 24 (bf16x3) MFMAs of iteration it | group 1 stages
//     phase 2*it + 1 : group 1 issues its MFMAs of iteration it             | group 0 stages
// "stages" = write the weight slab of iteration it+1 (registers, loaded one iteration ago) into the other LDS weight
// buffer, at the last tap write the next activation chunk into the other LDS activation buffer, then issue the
// global loads of the slab of iteration it+2 (and of the chunk after next).  Every load has a full iteration
// (>= 1500 cycles) to land, so one register set per operand is enough, and the staging group's waits, conversions and
// LDS writes all sit beside the other group's MFMAs.  The activation tile is double-buffered (the LDS budget of one
// workgroup per CU allows it), which also removes the chunk-boundary barrier of tapgemm.hip.
//
// Convolutions with N >= 128 only (ntaps >= 2, BK = 32, stride-1-sized halo); everything else stays on tapgemm.hip.
//
// STATUS: opt-in (SI_TG_PP=1).  Parity-green on the whole GPU suite, no spills (211 VGPRs), but 5-7 % SLOWER than
// tapgemm.hip's lockstep 8-wave tile on MI355X (bf16x3: 14.9 vs 14.0 ms/step; fp32: 38.4 vs 37.0).  In-kernel stamps
// (make stamps; tools/exp_stamps.py) put 37 % of a wave's life outside the loop (cold prologue 11 %, epilogue 19-26 %)
// and the two phases of an iteration at ~1100 cycles each against 768 cycles of MFMA issue.  Also tried on this
// kernel, without gain: staggering the first wave of workgroups by fractions of a tile time (the fixed per-tile cost
// is not chip-wide HBM contention), and 16-byte epilogue accesses via transposed accumulators (see the epilogue).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pp_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int MATH> struct PPElem { typedef float type; static constexpr int PAD = 4; };
template <> struct PPElem<SI_MATH_BF16> { typedef unsigned short type; static constexpr int PAD = 8; };
template <> struct PPElem<SI_MATH_BF16X3> { typedef unsigned short type; static constexpr int PAD = 8; };

// Diagnostic build only (make stamps, -DTG_STAMPS): s_memtime sums per wave group, si_pp_stamps[grp][8] =
// {compute phases, stage phases, barrier wait after compute, barrier wait after stage, prologue, epilogue, total, waves}.
#ifdef TG_STAMPS
__device__ unsigned long long si_pp_stamps[16];
#define PP_T(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define PP_ACC(slot, a, b) pst[slot] += (b) - (a)
extern "C" int si_debug_stamps_pp(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(si_pp_stamps), sizeof(si_pp_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(si_pp_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define PP_T(var)
#define PP_ACC(slot, a, b)
#endif

constexpr int PP_BM = 256, PP_BN = 128, PP_BK = 32, PP_NT = 512, PP_GT = 256, PP_MAXA = 10;

template <int MATH>
__global__ __launch_bounds__(PP_NT, 2) void tapgemm_pp_kernel(const TapGemmParams p) {
    constexpr int BM = PP_BM, BN = PP_BN, BK = PP_BK, GT = PP_GT, MAXA = PP_MAXA;
    constexpr int WARPS_N = 2;
    constexpr int WM = 64, WN = 64, TM = 2, TN = 2;
    typedef typename PPElem<MATH>::type elem_t;
    constexpr int LD = BK + PPElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    constexpr int V4 = BK / 4;
    constexpr int VB = (MATH == SI_MATH_F32) ? BK / 4 : BK / 8;
    constexpr int MAXB = BN * VB / GT;                             // 16-byte weight vectors per staging thread and plane
    static_assert(BN * VB % GT == 0, "every staging thread owns the same number of weight vectors");
    constexpr int KS = (MATH == SI_MATH_F32) ? BK / 8 : BK / 16;   // fragment loads per operand tile and iteration

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int gtid = tid & (GT - 1);                               // thread index inside its wave group
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);     // 0: waves 0-3 (rows 0..127), 1: waves 4-7
    // the second-dispatched half loses every issue arbitration against its SIMD partner (MI355X_MICROARCH.md, "Two
    // waves per SIMD", item 4): one static priority raise evens the two halves out
    if (grp) __builtin_amdgcn_s_setprio(1);

    const int mtiles = (p.M + BM - 1) / BM;
    const int ntn = (p.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn;
    const int seg = mt / mtiles;
    const int m0 = (mt % mtiles) * BM;
    const int n0 = (blockIdx.x % ntn) * BN;
    const int g = blockIdx.y;

    const int ntaps = p.ntaps;
    const int nchunks = p.Cin / BK;
    const int n_it = nchunks * ntaps;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int dil_lo = p.dil < 0 ? (ntaps - 1) * p.dil : 0;
    const int base_in = m0 * p.stride - p.pad + dil_lo;
    const int rowsA = (BM - 1) * p.stride + (ntaps - 1) * adil + 1;

    const size_t a_tile = (size_t)PLANES * rowsA * LD;
    constexpr size_t b_tile = (size_t)PLANES * BN * LD;
    elem_t* As = reinterpret_cast<elem_t*>(smem);                 // [2][PLANES][rowsA][LD]
    elem_t* Bs = As + 2 * a_tile;                                 // [2][PLANES][BN][LD]

    const size_t wplane = (size_t)ntaps * p.Npad * p.Cin;
    const float slope = p.pro_slope;
    const float* xs = p.x + (long)seg * p.x_seg_stride + (long)g * p.Cin;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging roles: group 0 owns the activation chunks (ra), group 1 the weight slabs (rb).  A wave's vmcnt is one
    // in-order counter, so a wave that staged both would wait for its (HBM-latency) activation loads every time it
    // needs its (L2-latency) weight loads; with the roles split each group's vmcnt(0) covers only its own stream.
    // One register array for both roles (a wave only ever plays one of them).
    static_assert(PLANES * MAXB <= MAXA, "the weight share fits the shared staging registers");
    f32x4 rs[MAXA];

    // Activation loads go through a buffer descriptor over this segment's input: rows before the segment (negative
    // offset = huge unsigned) and past its end fail the hardware range check and read as zero, so a slot is one
    // instruction with no predicate; a thread keeps one byte offset and adds a wave-uniform step per slot.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (int)((long)p.Lin * p.ldx * 4), 0x00020000);
    const int a_r0 = gtid / V4, a_j = gtid - a_r0 * V4;            // slot i covers tile row a_r0 + i * (GT / V4)
    const int a_voff = ((base_in + a_r0) * p.ldx + 4 * a_j) * 4;
    const int a_step = (GT / V4) * p.ldx * 4;
    auto issueA = [&](int chunk) {
        const int coff = chunk * BK * 4;
#pragma unroll
        for (int i = 0; i < MAXA; ++i)
            rs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, a_voff + i * a_step + coff, 0, 0));
    };
    auto storeA = [&](elem_t* Ad) {
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            const int r = a_r0 + i * (GT / V4), j = a_j;
            if (r < rowsA) {
                f32x4 v = rs[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                if constexpr (MATH == SI_MATH_F32) {
                    *reinterpret_cast<f32x4*>(Ad + r * LD + 4 * j) = v;
                } else {
                    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                    *reinterpret_cast<bf16x4*>(Ad + r * LD + 4 * j) = hi;
                    if constexpr (MATH == SI_MATH_BF16X3) {
                        const f32x4 rem = v - __builtin_convertvector(hi, f32x4);
                        *reinterpret_cast<bf16x4*>(Ad + (size_t)rowsA * LD + r * LD + 4 * j) = __builtin_convertvector(rem, bf16x4);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                     // one slot's temporaries at a time (register pressure)
        }
    };
    // Weight loads likewise: one descriptor per plane, one per-thread byte offset, the (chunk, tap, N-tile) part of the
    // address in the scalar offset.
    const int w_bytes = (int)(sizeof(elem_t) * (size_t)p.groups * wplane);
    const __amdgpu_buffer_rsrc_t wrsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PLANES == 2 ? p.w_lo : p.w), 0, w_bytes, 0x00020000);
    const int b_r0 = gtid / VB, b_j = gtid - b_r0 * VB;            // slot i covers slab row b_r0 + i * (GT / VB)
    const int b_voff = (int)sizeof(elem_t) * b_r0 * p.Cin + 16 * b_j;
    const int b_step = (int)sizeof(elem_t) * (GT / VB) * p.Cin;
    auto issueB = [&](int c, int t) {
        const int soff = (int)sizeof(elem_t) * (int)((size_t)g * wplane + ((size_t)t * p.Npad + n0) * p.Cin + c * BK);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < MAXB; ++i)
                rs[pl * MAXB + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pl == 0 ? wrsrc0 : wrsrc1, b_voff + i * b_step, soff, 0));
    };
    auto storeB = [&](elem_t* dst) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < MAXB; ++i)
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst + (size_t)pl * BN * LD + (b_r0 + i * (GT / VB)) * LD) + 16 * b_j) = rs[pl * MAXB + i];
    };

    // MFMA operand fragments of ONE iteration, loaded in the wave's staging phase and consumed in its compute phase:
    // the compute phase is then nothing but MFMAs (no LDS latency in front of the matrix pipe).
    f32x4 fa[PLANES][TM][KS], fb[PLANES][TN][KS];                  // 16-byte LDS vectors (fp32 x4 or bf16 x8)
    auto preload = [&](const elem_t* Ac, const elem_t* Bc, int tap) {
        const int toff = tap * p.dil - dil_lo;
        constexpr int HOFF = (MATH == SI_MATH_F32) ? BK / 2 : 8;   // element offset of the lane half
        constexpr int KOFF = (MATH == SI_MATH_F32) ? 4 : 16;       // elements per fragment step
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const elem_t* ap = Ac + (size_t)pl * rowsA * LD + ((wm0 + i * 32 + l31) * p.stride + toff) * LD + half * HOFF;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fa[pl][i][ks] = *reinterpret_cast<const f32x4*>(ap + KOFF * ks);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const elem_t* bp = Bc + (size_t)pl * BN * LD + (wn0 + j * 32 + l31) * LD + half * HOFF;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fb[pl][j][ks] = *reinterpret_cast<const f32x4*>(bp + KOFF * ks);
            }
        }
    };
    auto compute = [&]() {
        if constexpr (MATH == SI_MATH_F32) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][i][ks][e], fb[0][j][ks][e], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, fa[0][i][ks]), bh = __builtin_bit_cast(bf16x8, fb[0][j][ks]);
                        if constexpr (MATH == SI_MATH_BF16X3) {
                            const bf16x8 al = __builtin_bit_cast(bf16x8, fa[PLANES - 1][i][ks]), bl = __builtin_bit_cast(bf16x8, fb[PLANES - 1][j][ks]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                    }
        }
    };

#ifdef TG_STAMPS
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0};
#endif
    PP_T(t_begin);
    // ---- prologue: chunk 0 and slab 0 into LDS, chunk 1 and slab 1 in flight, group 0's first fragments loaded ----
    if (grp == 0) {
        issueA(0);
        storeA(As);
        if (nchunks > 1) issueA(1);
    } else {
        issueB(0, 0);
        storeB(Bs);
        issueB(0, 1);                                              // iteration 1 = (chunk 0, tap 1): ntaps >= 2
    }
    __syncthreads();
    PP_T(t_loop);
    PP_ACC(4, t_begin, t_loop);

    // ---- main loop.  Each group runs its OWN loop (the branch is wave-uniform and never changes), both execute two
    //      barriers per iteration:   group 0: [MFMAs of it] | [activation duties, fragments of it+1]
    //                                 group 1: [weight duties, fragments of it] | [MFMAs of it]
    //      One loop with a per-phase role test made the compiler keep two copies of every loop-carried register
    //      (fragments, staging registers) and spill. ----
    if (grp == 0) {
        preload(As, Bs, 0);
        int c = 0, t = 0;
#pragma clang loop unroll(disable)
        for (int it = 0; it < n_it; ++it) {
            PP_T(t_a);
            compute();
            PP_T(t_b);
            __syncthreads();
            PP_T(t_c);
            int c1 = c, t1 = t + 1;                                // iteration it + 1
            if (t1 == ntaps) { t1 = 0; ++c1; }
            // chunk c+1 goes to LDS one tap before its first use; the chunk after it is requested
            if (t == ntaps - 2 && c + 1 < nchunks) {
                storeA(As + (size_t)((c + 1) & 1) * a_tile);
                if (c + 2 < nchunks) issueA(c + 2);
            }
            if (it + 1 < n_it) preload(As + (size_t)(c1 & 1) * a_tile, Bs + (size_t)((it + 1) & 1) * b_tile, t1);
            PP_T(t_d);
            __syncthreads();
            PP_T(t_e);
            PP_ACC(0, t_a, t_b); PP_ACC(2, t_b, t_c); PP_ACC(1, t_c, t_d); PP_ACC(3, t_d, t_e);
            c = c1; t = t1;
        }
    } else {
        int c = 0, t = 0;
#pragma clang loop unroll(disable)
        for (int it = 0; it < n_it; ++it) {
            PP_T(t_a);
            int c1 = c, t1 = t + 1;
            if (t1 == ntaps) { t1 = 0; ++c1; }
            // slab it+1 (requested one iteration ago) goes to LDS, slab it+2 is requested
            if (it + 1 < n_it) storeB(Bs + (size_t)((it + 1) & 1) * b_tile);
            if (it + 2 < n_it) {
                int c2 = c1, t2 = t1 + 1;
                if (t2 == ntaps) { t2 = 0; ++c2; }
                issueB(c2, t2);
            }
            preload(As + (size_t)(c & 1) * a_tile, Bs + (size_t)(it & 1) * b_tile, t);
            PP_T(t_b);
            __syncthreads();
            PP_T(t_c);
            compute();
            PP_T(t_d);
            __syncthreads();
            PP_T(t_e);
            PP_ACC(1, t_a, t_b); PP_ACC(3, t_b, t_c); PP_ACC(0, t_c, t_d); PP_ACC(2, t_d, t_e);
            c = c1; t = t1;
        }
    }
    PP_T(t_epi);

    // ---- epilogue (as tapgemm.hip: one burst of residual / accumulate loads through buffer descriptors, then all
    //      stores).  A variant with the MFMA operands swapped (transposed accumulators: each lane owns runs of four
    //      consecutive n, so every access is 16 bytes wide, 16 instead of 64 per tile) measured 6 % SLOWER: rows of
    //      128 contiguous bytes per wave-instruction beat 32 rows x 32 bytes. ----
    float* const outp = p.out + (long)seg * p.o_seg_stride;
    const float* const resp = p.res ? p.res + (long)seg * p.o_seg_stride : outp;
    const bool has_res = p.res != nullptr;
    const bool acc_out = p.accumulate != 0;
    const bool gelu = p.act == SI_ACT_GELU;
    const int nbytes = (int)p.olimit * 4;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resp), 0, nbytes, 0x00020000);
    const int rstep = p.ldo * 4;                                   // bytes between output rows
    int vb[TM][TN];
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + l31;
        const bool nok = n < p.N;
        bv[j] = (p.bias && nok) ? p.bias[g * p.N + n] : 0.f;
        const int col = g * p.N + n + (int)p.ooff;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            vb[i][j] = nok ? ((m0 + wm0 + i * 32 + 4 * half) * p.ldo + col) * 4 : (int)0x80000000;
    }
    // C/D row of accumulator register r (besides the 4*half already in vb): (r&3) + 8*(r>>2)
    float rv[TM][TN][16], ov[TM][TN][16];
    if (has_res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, vb[i][j] + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0));
    }
    if (acc_out) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ov[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(orsrc, vb[i][j] + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv[j];
                if (gelu) v = pp_gelu_erf(v);
                if (has_res) v += rv[i][j][r];
                v *= p.alpha;
                if (acc_out) v += ov[i][j][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, vb[i][j] + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0);
            }
#ifdef TG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_T(t_end);
    PP_ACC(5, t_epi, t_end);
    if (lane == 0) {
        for (int q = 0; q < 6; ++q) atomicAdd(&si_pp_stamps[grp * 8 + q], pst[q]);
        atomicAdd(&si_pp_stamps[grp * 8 + 6], t_end - t_begin);
        atomicAdd(&si_pp_stamps[grp * 8 + 7], 1ull);
    }
#endif
}

template <int MATH>
static int pp_launch(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    typedef typename PPElem<MATH>::type elem_t;
    constexpr int LD = PP_BK + PPElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int rowsA = (PP_BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const size_t lds = (size_t)PLANES * (2 * (size_t)rowsA + 2 * PP_BN) * LD * sizeof(elem_t);
    if (rowsA * (PP_BK / 4) > PP_MAXA * PP_GT || lds > 160 * 1024) return 1;
    auto kern = tapgemm_pp_kernel<MATH>;
    static size_t lds_set = 0;
    if (lds > 64 * 1024 && lds > lds_set) {
        SI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    const int mtiles = (p.M + PP_BM - 1) / PP_BM;
    dim3 grid((unsigned)(p.nseg * mtiles * ((p.N + PP_BN - 1) / PP_BN)), (unsigned)p.groups);
    static const char* const math_names[] = {"f32", "bf16", "bf16x3"};
    char name[48];
    snprintf(name, sizeof(name), "tapgemm_pp_%s_256x128", math_names[MATH]);
    const double macs = p.algo_macs > 0 ? p.algo_macs : (double)p.nseg * p.M * p.N * p.groups * (double)p.Cin * p.ntaps;
    double bytes = 4.0 * p.nseg * ((double)p.Lin * p.Cin * p.groups + (double)p.M * p.N * p.groups * (1 + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0))) +
                   (double)p.groups * p.ntaps * p.N * p.Cin * (MATH == SI_MATH_F32 ? 4 : (MATH == SI_MATH_BF16 ? 2 : 4));
    si_prof_begin(ctx, name, 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(kern, grid, dim3(PP_NT), lds, st, p);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller falls back to tapgemm.hip).
int si_launch_tapgemm_pp(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st) {
    if (p.ntaps < 2 || p.Cin % 32 != 0 || p.N < 128 || p.M <= 256 || p.groups != 1 || p.x16 || p.out16 || !p.out) return 1;
    if ((long)p.Lin * p.ldx * 4 >= (1L << 31)) return 1;           // 32-bit byte offsets into the segment's input
    switch (math) {
        case SI_MATH_F32: return pp_launch<SI_MATH_F32>(ctx, p, st);
        case SI_MATH_BF16: return pp_launch<SI_MATH_BF16>(ctx, p, st);
        case SI_MATH_BF16X3: return pp_launch<SI_MATH_BF16X3>(ctx, p, st);
    }
    return 1;
}
