// tapgemm.hip -- the contraction kernel of the path (gfx950, wave64, MFMA).
//
// Computes, on channels-last fp32 activations,
//     out[seg][m][n] = epi( sum_tap sum_ci pro(x[seg][m*stride + tap*dil - pad][g*Cin + ci]) * W[g][tap][n][ci] )
// which covers Conv1d (stride / dilation / groups / zero padding), Linear (ntaps = 1) and ConvTranspose1d
// (split into its `stride` output phases: 2 taps with dil = -1, N = stride*Cout, see api.hip).
// Replaces the torch.nn calls of SURVEY.md 8(a) rows A2-A9, B1-B3.
//
// Structure: one workgroup of 4 or 8 waves owns a BM x BN output tile (the launcher at the bottom picks the shape per
// layer; DESIGN.md 4.1 has the table and the measurements behind each choice).  For every K chunk of BK input
// channels the halo'd activation tile ((BM-1)*stride + (ntaps-1)*|dil| + 1 rows) is staged ONCE into LDS and
// reused by all taps -- a tap is just a row offset into that tile -- while the per-tap BK x BN weight slab
// (L2-resident, shared by every workgroup) is re-staged per tap.  Activations come either as fp32 (prologue
// leaky-relu and the fp32 -> bf16 / fp16 / bf16 hi+lo conversion happen while staging) or "operand-ready" as 16-bit
// values already in the MFMA operand type (template flag A16: staging is a copy); the epilogue (bias, GELU, residual,
// scale, accumulate, optional 16-bit copy for the consumer) works on the accumulators.
//
// Pipeline: the loop runs over (chunk, tap) iterations.  In the 16-bit modes an iteration's MFMA work is SHORTER than
// the L2 latency of a weight slab, so slabs are prefetched TWO iterations ahead into two register sets; the next
// activation chunk is issued at the first tap of the current chunk, or two chunks ahead in two half-sets for
// ntaps == 1 (Linear).  Registers are written to the OTHER LDS weight buffer after the MFMAs: one workgroup barrier
// per iteration (plus one per chunk boundary for convolutions, whose activation tile is single-buffered).  All
// operand loads go through buffer descriptors (hardware range check instead of predicates) and are issued
// unconditionally wherever possible, so the compiler's counted s_waitcnt vmcnt(N) leaves the younger batches in flight.
//
// MFMA use (cdna_hip_programming.md section 3):
//   F32    v_mfma_f32_32x32x2_f32 : lane l supplies A[l&31][k=l>>5], B[k=l>>5][l&31].  The k index is a
//          dummy, so lane-half h is given the K range [h*BK/2, (h+1)*BK/2) of the chunk: consecutive MFMA
//          steps then read consecutive floats and one ds_read_b128 feeds four steps.
//   BF16   v_mfma_f32_32x32x16_bf16: lane supplies 8 consecutive k (16 B) of row l&31, k block l>>5.
//   BF16X3 same instruction three times (lo*hi + hi*lo + hi*hi) for ~fp32 accuracy at 3/16 of the fp32 cost.
//   F16    v_mfma_f32_32x32x16_f16, same fragment layout; operands saturate at +-65504.
// LDS rows are padded by 16 B so the 16 rows a ds_read_b128 lane group touches fall on distinct 4-bank slots.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int MATH> struct LdsElem { typedef float type; static constexpr int PAD = 4; };
template <> struct LdsElem<SI_MATH_BF16> { typedef unsigned short type; static constexpr int PAD = 8; };
template <> struct LdsElem<SI_MATH_BF16X3> { typedef unsigned short type; static constexpr int PAD = 8; };
template <> struct LdsElem<SI_MATH_F16> { typedef unsigned short type; static constexpr int PAD = 8; };

// float4 of the activation tile a thread holds in flight: 10 covers 128-row tiles (<= 320 rows at BK = 32),
// 12 the 256-row tiles (306 rows) and the positional conv (383 rows at BK = 16)
// (an 8-wave workgroup spreads the same tile over 512 threads: half the registers per thread)
template <int BM, int NT = 256, int BK = 32> struct MaxA { static constexpr int value = BK == 64 ? 8 : (NT == 512 ? 6 : (BM == 128 ? 10 : 12)); };

template <int V> using ic = std::integral_constant<int, V>;

// Diagnostic build only (make stamps -> libsi_hip_stamps.so, -DTG_STAMPS): per-phase s_memtime sums of every wave,
// accumulated per N-tile family into si_tg_stamps[3][8] = {issue, compute, land (vmcnt wait + LDS writes), barrier,
// prologue, epilogue, total, waves}; read with tools/exp_stamps.py.  The stamps serialise the phases they bracket, so
// only the SHARES are meaningful; no stamp executes in the shipped library.
#ifdef TG_STAMPS
__device__ unsigned long long si_tg_stamps[24];
#define TG_T(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define TG_ACC(slot, a, b) st_acc[slot] += (b) - (a)
#else
#define TG_T(var)
#define TG_ACC(slot, a, b)
#endif

// LINEAR (ntaps == 1) is a compile-time variant so that each instantiation carries only its own loop and register sets.
// __launch_bounds__(256, 2): two waves per SIMD = two workgroups per CU (what the LDS footprint allows); without the
// second argument the allocator takes up to 235 VGPRs + 64 accumulators and halves the occupancy.
// A16: the activations arrive operand-ready (p.x16: 16-bit, already in the MFMA operand type of MATH, prologue
// activation already applied by the producer's epilogue): staging is an 8-byte copy per 4 channels, no conversion.
// Occupancy target: 2 waves per SIMD in general.  The operand-ready convolution kernels are light enough for 4 (128 VGPRs:
// half-width staging registers, per-tile epilogue): two 8-wave / four 4-wave workgroups per CU, so one workgroup's cold
// prologue and epilogue overlap another's main loop (measured: 256x128 8.8 -> 7.8 ms/step, 256x32 3.3 -> 3.1).  The
// 4-wave 256x64 and 128x128 tiles spill at 128 registers and lose (4.1 -> 5.0), Linear layers likewise: they stay at 2.
template <int BM, int BN, int NT, bool LINEAR, bool A16> struct WavesPerSimd { static constexpr int value = (A16 && !LINEAR && !(BM == 256 && BN == 64 && NT == 256) && !(BM == 128 && BN == 128 && NT == 256)) ? 4 : 2; };

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK, bool LINEAR, bool A16 = false>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, (WavesPerSimd<BM, BN, 64 * WARPS_M * WARPS_N, LINEAR, A16>::value)) void tapgemm_kernel(const TapGemmParams p) {
    static_assert(!A16 || MATH == SI_MATH_BF16 || MATH == SI_MATH_F16, "operand-ready activations are 16-bit single-plane");
    static_assert(WARPS_M * WARPS_N == 4 || WARPS_M * WARPS_N == 8, "4 or 8 waves per workgroup");
    constexpr int NT = 64 * WARPS_M * WARPS_N;          // threads per workgroup
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile is a multiple of 32x32");
    typedef typename LdsElem<MATH>::type elem_t;
    constexpr int LD = BK + LdsElem<MATH>::PAD;          // LDS row stride in elements (row = BK*sizeof + 16 B)
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    constexpr int V4 = BK / 4;                           // float4 per activation row
    constexpr int VB = (MATH == SI_MATH_F32) ? BK / 4 : BK / 8;     // 16-byte vectors per weight row (per plane)
    constexpr int MAXB = (BN * VB + NT - 1) / NT;          // 16-byte vectors of a weight slab per thread (per plane)
    constexpr int MAXA = MaxA<BM, NT, BK>::value;
    constexpr int HALF = MAXA / 2;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;

    const int mtiles = (p.M + BM - 1) / BM;
    // N-tiles of one M-tile are adjacent in dispatch order: they read the same activation rows, so all but the
    // first find them in L2 / Infinity Cache instead of HBM.
    const int ntn = (p.N + BN - 1) / BN;
    const int tile = blockIdx.x;
    const int mt = tile / ntn;
    const int seg = mt / mtiles;
    const int m0 = (mt % mtiles) * BM;
    const int n0 = (tile % ntn) * BN;
    const int g = blockIdx.y;
    // ragged batches: this segment's own row counts (wave-uniform scalar loads); a tile past its rows has nothing to do
    const int Lin_s = p.seg_lin ? p.seg_lin[seg] : p.Lin;
    if (p.seg_m && m0 >= p.seg_m[seg]) return;
    const long olimit_s = p.seg_orows ? (long)p.seg_orows[seg] * p.olim_mul : p.olimit;
    const long xseg0 = p.seg_row_off ? (long)p.seg_row_off[seg] * p.ldx : (long)seg * p.x_seg_stride;
    const long oseg0 = p.seg_row_off ? (long)p.seg_row_off[seg] * p.ldo : (long)seg * p.o_seg_stride;

    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int dil_lo = p.dil < 0 ? (p.ntaps - 1) * p.dil : 0;
    const int base_in = m0 * p.stride - p.pad + dil_lo;          // input row held in LDS row 0
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const int ntaps = p.ntaps;
    constexpr bool linear = LINEAR;
    const int abufs = linear ? 2 : 1;

    const size_t a_tile = (size_t)PLANES * rowsA * LD;            // elements per activation buffer
    constexpr size_t b_tile = (size_t)PLANES * BN * LD;           // elements per weight buffer
    elem_t* As = reinterpret_cast<elem_t*>(smem);                 // [abufs][PLANES][rowsA][LD]
    elem_t* Bs = As + (size_t)abufs * a_tile;                     // [2][PLANES][BN][LD]

    const float* xs = A16 ? nullptr : p.x + xseg0 + (long)g * p.Cin;
    const unsigned short* xs16 = A16 ? p.x16 + xseg0 + (long)g * p.Cin : nullptr;
    const size_t wplane = (size_t)ntaps * p.Npad * p.Cin;        // elements per group
    const float slope = p.pro_slope;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef TG_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    TG_T(st_begin);
#endif
    typedef typename std::conditional<A16, f32x2, f32x4>::type ra_t;   // 4 channels per slot: 8 bytes operand-ready, 16 bytes fp32
    ra_t ra[MAXA];                               // activation chunk(s) in flight: one set, or two half-sets (Linear)
    f32x4 rb0[PLANES][MAXB], rb1[PLANES][MAXB];  // weight slabs of iterations it+1 / it+2 in flight

    // Operand loads go through buffer descriptors: a thread keeps ONE byte offset per operand and adds a compile-time
    // multiple of a wave-uniform step per slot; rows before / after the segment (negative offset = huge unsigned, or
    // >= num_records) fail the hardware range check and read as zero.  No 64-bit address registers, no zero-init and no
    // branch per slot -- with pointer loads the compiler spilled addresses and drained vmcnt(0) around the reloads and
    // around every predicated slot, several times per K chunk.
    constexpr int AESZ = A16 ? 2 : 4;
    const void* const xbase = A16 ? static_cast<const void*>(xs16) : static_cast<const void*>(xs);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(xbase), 0, (int)(((long)Lin_s * p.ldx - (long)g * p.Cin) * AESZ), 0x00020000);
    static_assert(NT % V4 == 0, "slot i of a thread is row r0 + i * (NT / V4), same column group");
    const int a_r0 = tid / V4, a_j = tid - a_r0 * V4;
    const int a_voff = ((base_in + a_r0) * p.ldx + 4 * a_j) * AESZ;
    const int a_step = (NT / V4) * p.ldx * AESZ;
    // registers [LO, LO+CNT) of `ra` <- chunk c0 of the activation tile (zero outside the clip)
    auto issueA = [&](auto lo, auto cnt, int c0) {
        constexpr int LO = decltype(lo)::value, CNT = decltype(cnt)::value;
        const int coff = c0 * AESZ;
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            if constexpr (A16) ra[LO + i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, a_voff + i * a_step + coff, 0, 0));
            else ra[LO + i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, a_voff + i * a_step + coff, 0, 0));
        }
    };
    auto storeA = [&](auto lo, auto cnt, elem_t* dst) {
        constexpr int LO = decltype(lo)::value, CNT = decltype(cnt)::value;
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int r = a_r0 + i * (NT / V4), j = a_j;
            if (r < rowsA) {
                if constexpr (A16) {
                    f32x2 raw = ra[LO + i];
                    if (slope != 1.f) {          // raw 16-bit activation stream: the prologue activation on packed values
                        if constexpr (MATH == SI_MATH_F16) {
                            f16x4 h = __builtin_bit_cast(f16x4, raw);
                            const f16x4 hs = h * (_Float16)slope;
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = h[e] > (_Float16)0 ? h[e] : hs[e];
                            raw = __builtin_bit_cast(f32x2, h);
                        } else {
                            f32x4 f = __builtin_convertvector(__builtin_bit_cast(bf16x4, raw), f32x4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * slope;
                            raw = __builtin_bit_cast(f32x2, __builtin_convertvector(f, bf16x4));
                        }
                    }
                    *reinterpret_cast<f32x2*>(dst + r * LD + 4 * j) = raw;
                    continue;
                }
                f32x4 v;
                if constexpr (!A16) v = ra[LO + i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                if constexpr (MATH == SI_MATH_F32) {
                    *reinterpret_cast<f32x4*>(dst + r * LD + 4 * j) = v;
                } else if constexpr (MATH == SI_MATH_F16) {
                    // fp16 operands (11-bit significand): saturate instead of overflowing to infinity
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fminf(__builtin_fmaxf(v[e], -65504.f), 65504.f);
                    *reinterpret_cast<f16x4*>(dst + r * LD + 4 * j) = __builtin_convertvector(v, f16x4);
                } else {
                    // vector casts lower to v_cvt_pk_bf16_f32 (round-to-nearest-even, 2 elements per instruction)
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
    const int w_bytes = (int)(sizeof(elem_t) * (size_t)p.groups * wplane);
    const __amdgpu_buffer_rsrc_t wrsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PLANES == 2 ? p.w_lo : p.w), 0, w_bytes, 0x00020000);
    static_assert(NT % VB == 0, "slot i of a thread is slab row r0 + i * (NT / VB), same 16-byte column");
    const int b_r0 = tid / VB, b_j = tid - b_r0 * VB;
    const int b_step = (int)sizeof(elem_t) * (NT / VB) * p.Cin;
    // threads beyond the slab (narrow tiles: BN * VB < NT) load out of range (zero, never stored)
    const int b_voff = (BN * VB % NT == 0 || tid < BN * VB) ? (int)sizeof(elem_t) * b_r0 * p.Cin + 16 * b_j : (int)0x80000000;
    auto issueB = [&](f32x4 (&rb)[PLANES][MAXB], int c0, int tap) {
        const int soff = (int)(sizeof(elem_t) * ((size_t)g * wplane + ((size_t)tap * p.Npad + n0) * p.Cin + c0));
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < MAXB; ++i)
                rb[pl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pl == 0 ? wrsrc0 : wrsrc1, b_voff + i * b_step, soff, 0));
    };
    auto storeB = [&](const f32x4 (&rb)[PLANES][MAXB], elem_t* dst) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int r = b_r0 + i * (NT / VB);
                if (BN * VB % NT == 0 || r < BN)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst + (size_t)pl * BN * LD + r * LD) + 16 * b_j) = rb[pl][i];
            }
    };

    // ---- MFMA over one (chunk, tap) out of LDS ----
    auto compute = [&](const elem_t* Ac, const elem_t* Bc, int tap) {
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
                    for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const bf16x8*>(ap[i] + (size_t)rowsA * LD + 16 * ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(bp[j] + (size_t)BN * LD + 16 * ks);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            // small terms first so they are not swamped by the running sum
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                } else if constexpr (MATH == SI_MATH_F16) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bh[j]), acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    const int nchunks = p.Cin / BK;
    const int total = nchunks * ntaps;
    // (chunk, tap) of iterations it, it+1, it+2
    int c0 = 0, t0 = 0, c1 = 0, t1 = 0, c2 = 0, t2 = 0;
    auto adv = [&](int& c, int& t) { if (++t == ntaps) { t = 0; ++c; } };
    adv(c1, t1);
    adv(c2, t2); adv(c2, t2);

    // (Measured and rejected: delaying the second resident workgroup of each CU by 1/2 or 1 tile to break the
    // lock-step of co-resident workgroups cost 3-5 % on every fp32 shape.)
    if constexpr (LINEAR) {
        // ================================================================== Linear: every iteration is a new chunk
        // Prefetch loads are UNCONDITIONAL (descriptor clamped to the last chunk, which is simply re-read): a load under
        // `if (it + 2 < total)` makes the compiler's s_waitcnt conservative at the branch join -- it must assume the
        // load was not issued, emits vmcnt(0), and so drains the very prefetch that was meant to stay in flight (an
        // ablation showed each iteration paying one full memory latency).
        issueA(ic<0>{}, ic<HALF>{}, 0);
        issueB(rb0, 0, 0);
        issueA(ic<HALF>{}, ic<HALF>{}, total > 1 ? BK : 0);
        issueB(rb1, total > 1 ? BK : 0, 0);
        storeA(ic<0>{}, ic<HALF>{}, As);
        storeB(rb0, Bs);
        __syncthreads();
        // even step: set 0 (ra[0..HALF), rb0) receives iteration it+2, set 1 lands iteration it+1; odd step mirrored
        auto step = [&](auto issue_lo, auto land_lo, f32x4 (&rissue)[PLANES][MAXB], const f32x4 (&rland)[PLANES][MAXB], int it) {
            const int cn = (it + 2 < total ? it + 2 : total - 1) * BK;
            issueB(rissue, cn, 0);
            issueA(issue_lo, ic<HALF>{}, cn);
            compute(As + (size_t)(it & 1) * a_tile, Bs + (size_t)(it & 1) * b_tile, 0);
            if (it + 1 < total) {
                storeB(rland, Bs + (size_t)((it + 1) & 1) * b_tile);
                storeA(land_lo, ic<HALF>{}, As + (size_t)((it + 1) & 1) * a_tile);
                __syncthreads();
            }
        };
        for (int it = 0; it < total; it += 2) {
            step(ic<0>{}, ic<HALF>{}, rb0, rb1, it);
            if (it + 1 < total) step(ic<HALF>{}, ic<0>{}, rb1, rb0, it + 1);
        }
    } else if constexpr (MATH == SI_MATH_BF16X3 && BN < 128) {
        // ================================================================== convolution, weights ONE iteration ahead:
        // on the narrow bf16x3 tiles the second register set pushed the kernel to one wave per SIMD (193 VGPRs + 64
        // accumulators) and cost 45 %; these tiles are HBM-bound anyway
        issueA(ic<0>{}, ic<MAXA>{}, 0);
        issueB(rb0, 0, 0);
        storeA(ic<0>{}, ic<MAXA>{}, As);
        storeB(rb0, Bs);
        __syncthreads();
        TG_T(st_pro);
        TG_ACC(4, st_begin, st_pro);
        for (int it = 0; it < total; ++it) {
            const bool has_next = it + 1 < total;
            TG_T(s0);
            issueB(rb0, has_next ? c1 * BK : c0 * BK, has_next ? t1 : t0);      // unconditional (see the Linear loop)
            if (t0 == 0 && c0 + 1 < nchunks) issueA(ic<0>{}, ic<MAXA>{}, (c0 + 1) * BK);
            TG_T(s1);
            compute(As, Bs + (size_t)(it & 1) * b_tile, t0);
            TG_T(s2);
            TG_ACC(0, s0, s1); TG_ACC(1, s1, s2);
            if (has_next) {
                const bool new_chunk = t1 == 0;
                if (new_chunk) __syncthreads();
                storeB(rb0, Bs + (size_t)((it + 1) & 1) * b_tile);
                if (new_chunk) storeA(ic<0>{}, ic<MAXA>{}, As);
                TG_T(s3);
                __syncthreads();
                TG_T(s4);
                TG_ACC(2, s2, s3); TG_ACC(3, s3, s4);
            }
            c0 = c1; t0 = t1;
            adv(c1, t1);
        }
    } else {
        // ================================================================== convolution: ntaps iterations per chunk
        // The weight slab of iteration it+2 is issued UNCONDITIONALLY (descriptor clamped to the last iteration), see
        // the Linear loop for why.  The once-per-chunk activation prefetch stays conditional: at its join the compiler
        // settles for the smaller count, which drains part of that prefetch once per `ntaps` iterations, not every one.
        // (Four straight-line copies of the body -- chunk-opening x register-set parity -- were exact but spilled.)
        if (total == 1) { c1 = c0; t1 = t0; }
        if (total <= 2) { c2 = c1; t2 = t1; }
        issueA(ic<0>{}, ic<MAXA>{}, 0);
        issueB(rb0, 0, 0);
        issueB(rb1, c1 * BK, t1);
        storeA(ic<0>{}, ic<MAXA>{}, As);
        storeB(rb0, Bs);
        __syncthreads();
        TG_T(st_pro);
        TG_ACC(4, st_begin, st_pro);
        auto step = [&](f32x4 (&rissue)[PLANES][MAXB], const f32x4 (&rland)[PLANES][MAXB], int it) {
            TG_T(s0);
            issueB(rissue, c2 * BK, t2);
            if (t0 == 0 && c0 + 1 < nchunks) issueA(ic<0>{}, ic<MAXA>{}, (c0 + 1) * BK);   // lands over the chunk's other taps
            TG_T(s1);
            compute(As, Bs + (size_t)(it & 1) * b_tile, t0);
            TG_T(s2);
            TG_ACC(0, s0, s1); TG_ACC(1, s1, s2);
            if (it + 1 < total) {
                const bool new_chunk = t1 == 0;
                if (new_chunk) __syncthreads();                    // every wave is done reading the activation tile
                storeB(rland, Bs + (size_t)((it + 1) & 1) * b_tile);
                if (new_chunk) storeA(ic<0>{}, ic<MAXA>{}, As);
                TG_T(s3);
                __syncthreads();
                TG_T(s4);
                TG_ACC(2, s2, s3); TG_ACC(3, s3, s4);
            }
            c0 = c1; t0 = t1; c1 = c2; t1 = t2;
            if (it + 3 < total) adv(c2, t2);                       // the far descriptor stops at the last iteration
        };
        for (int it = 0; it < total; it += 2) {
            step(rb0, rb1, it);
            if (it + 1 < total) step(rb1, rb0, it + 1);
        }
    }

    TG_T(st_epi);
    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    // In-kernel stamps showed the epilogue at 31-47 % of a wave's life in the bf16 modes: eight dependent
    // "8 loads -> wait -> 8 stores" round trips per wave.  Now EVERY residual / accumulate read of the wave tile is
    // issued in one burst, then one wait, then all stores.  Accesses go through buffer descriptors that span exactly
    // this segment's output: the hardware range check drops rows >= M, the negative offsets of the ConvTranspose
    // phase layout and (voffset forced to 2^31) the columns >= N, so no per-element predicate or address register
    // survives -- a lane keeps one byte offset per 32x32 tile and adds a scalar row step.
    // p.out may be NULL when only the operand-ready 16-bit copy is wanted (p.out16); the descriptor then has 0 records
    float* const outp = p.out ? p.out + oseg0 : nullptr;
    const float* const resp = p.res ? p.res + oseg0 : outp;
    const bool has_res = p.res != nullptr || p.res16 != nullptr;
    const bool res_is16 = p.res16 != nullptr;
    const bool acc_out = p.accumulate != 0;
    const bool acc_is16 = p.acc16 != 0;
    const bool has_out = p.out != nullptr;
    const bool has_o16 = p.out16 != nullptr;
    const float slope16 = p.out16_slope;
    const bool gelu = p.act == SI_ACT_GELU;
    const int nbytes = (int)olimit_s * 4;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, has_out ? nbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resp), 0, resp ? nbytes : 0, 0x00020000);
    unsigned short* const o16p = has_o16 ? p.out16 + oseg0 : nullptr;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(o16p, 0, has_o16 ? nbytes / 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r16rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(res_is16 ? p.res16 + oseg0 : nullptr), 0, res_is16 ? nbytes / 2 : 0, 0x00020000);
    auto from16 = [](unsigned short h) -> float {
        if constexpr (MATH == SI_MATH_F16) return (float)__builtin_bit_cast(_Float16, h);
        else return __builtin_bit_cast(float, (unsigned)h << 16);
    };
    // four packed 16-bit values of the math mode's operand type -> fp32
    auto from16x4 = [](u32x2 pk) -> f32x4 {
        if constexpr (MATH == SI_MATH_F16) return __builtin_convertvector(__builtin_bit_cast(f16x4, pk), f32x4);
        else return __builtin_convertvector(__builtin_bit_cast(bf16x4, pk), f32x4);
    };
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
    constexpr bool light = WavesPerSimd<BM, BN, NT, LINEAR, A16>::value == 4;
    // Row-contiguous epilogue (the 4-waves-per-SIMD kernels, when columns come in aligned groups of four): each 32x32
    // accumulator tile is transposed through a private 4.5 KB LDS patch so that a lane owns four CONSECUTIVE columns of
    // one output row -- residual / accumulate reads and stores are 16 bytes per lane, 8 lanes per 128-byte row segment,
    // 4 wave-instructions per tile and stream instead of 16 per-lane scalars (and 8-byte stores for the 16-bit copy).
    if constexpr (light) {
        if (p.wide_epilogue) {
            __syncthreads();                                       // every wave is done with the operand tiles in LDS
            float* const tl = reinterpret_cast<float*>(smem) + wave * (32 * 36);
            const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn0 + j * 32 + lc;          // first of this lane's four columns
                    const bool nok = n < p.N;
                    const f32x4 b4 = (p.bias && nok) ? *reinterpret_cast<const f32x4*>(p.bias + g * p.N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; ++r) tl[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = acc[i][j][r];
                    const int obase = nok ? ((m0 + wm0 + i * 32 + lr) * p.ldo + g * p.N + n + (int)p.ooff) * 4 : (int)0x80000000;
#pragma unroll
                    for (int q0 = 0; q0 < 4; q0 += 2) {
                        f32x4 a[2], rr[2], oo[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int off = nok ? obase + (q0 + u) * 8 * rstep : (int)0x80000000;
                            a[u] = *reinterpret_cast<const f32x4*>(tl + (lr + 8 * (q0 + u)) * 36 + lc);
                            const int off16 = nok ? off / 2 : (int)0x80000000;
                            if (has_res) rr[u] = res_is16 ? from16x4(__builtin_amdgcn_raw_buffer_load_b64(r16rsrc, off16, 0, 0))
                                                          : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0));
                            if (acc_out) oo[u] = acc_is16 ? from16x4(__builtin_amdgcn_raw_buffer_load_b64(hrsrc, off16, 0, 0))
                                                          : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, off, 0, 0));
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int off = nok ? obase + (q0 + u) * 8 * rstep : (int)0x80000000;
                            f32x4 v;
                            f16x4 h16;
                            bf16x4 b16;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = a[u][e] + b4[e];
                                if (gelu) x = gelu_erf(x);
                                if (has_res) x += rr[u][e];
                                x *= p.alpha;
                                if (acc_out) x += oo[u][e];
                                v[e] = x;
                                float w = x > 0.f ? x : x * slope16;
                                if constexpr (MATH == SI_MATH_F16) h16[e] = (_Float16)__builtin_fminf(__builtin_fmaxf(w, -65504.f), 65504.f);
                                else b16[e] = (__bf16)w;
                            }
                            if (has_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
                            if (has_o16) {
                                const u32x2 pk = MATH == SI_MATH_F16 ? __builtin_bit_cast(u32x2, h16) : __builtin_bit_cast(u32x2, b16);
                                __builtin_amdgcn_raw_buffer_store_b64(pk, hrsrc, nok ? off / 2 : (int)0x80000000, 0, 0);
                            }
                        }
                    }
                }
            return;
        }
    }
    // C/D row of accumulator register r (besides the 4*half already in vb): (r&3) + 8*(r>>2)
    // The burst covers the whole wave tile, except in the kernels built for four waves per SIMD (128 VGPRs), which go
    // through it one 32x32 tile at a time: their co-resident waves cover the round trips.
    constexpr int IG = light ? 1 : TM, JG = light ? 1 : TN;
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += IG)
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += JG) {
        float rv[IG][JG][16], ov[IG][JG][16];
        if (has_res) {
#pragma unroll
            for (int i = 0; i < IG; ++i)
#pragma unroll
                for (int j = 0; j < JG; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = vb[i0 + i][j0 + j] + ((r & 3) + 8 * (r >> 2)) * rstep;
                        if (res_is16) rv[i][j][r] = from16(__builtin_amdgcn_raw_buffer_load_b16(r16rsrc, vb[i0 + i][j0 + j] == (int)0x80000000 ? (int)0x80000000 : off / 2, 0, 0));
                        else rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, off, 0, 0));
                    }
        }
        if (acc_out) {
#pragma unroll
            for (int i = 0; i < IG; ++i)
#pragma unroll
                for (int j = 0; j < JG; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = vb[i0 + i][j0 + j] + ((r & 3) + 8 * (r >> 2)) * rstep;
                        if (acc_is16) ov[i][j][r] = from16(__builtin_amdgcn_raw_buffer_load_b16(hrsrc, vb[i0 + i][j0 + j] == (int)0x80000000 ? (int)0x80000000 : off / 2, 0, 0));
                        else ov[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(orsrc, off, 0, 0));
                    }
        }
#pragma unroll
        for (int ii = 0; ii < IG; ++ii)
#pragma unroll
            for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = i0 + ii, j = j0 + jj;
                    float v = acc[i][j][r] + bv[j];
                    if (gelu) v = gelu_erf(v);
                    if (has_res) v += rv[ii][jj][r];
                    v *= p.alpha;
                    if (acc_out) v += ov[ii][jj][r];
                    const int off = vb[i][j] + ((r & 3) + 8 * (r >> 2)) * rstep;
                    if (has_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, off, 0, 0);
                    if constexpr (MATH == SI_MATH_BF16 || MATH == SI_MATH_F16) {
                        if (has_o16) {  // operand-ready copy for the consumer: its prologue activation, then its operand rounding
                            float w = v > 0.f ? v : v * slope16;
                            unsigned short h;
                            if constexpr (MATH == SI_MATH_F16) {
                                w = __builtin_fminf(__builtin_fmaxf(w, -65504.f), 65504.f);
                                h = __builtin_bit_cast(unsigned short, (_Float16)w);
                            } else {
                                h = __builtin_bit_cast(unsigned short, (__bf16)w);
                            }
                            // a masked column carries offset 2^31: halving it would bring it back into range
                            __builtin_amdgcn_raw_buffer_store_b16(h, hrsrc, vb[i][j] == (int)0x80000000 ? (int)0x80000000 : off / 2, 0, 0);
                        }
                    }
                }
    }
#ifdef TG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // include the store drain: a wave cannot retire before it
    TG_T(st_end);
    TG_ACC(5, st_epi, st_end);
    if (lane == 0) {
        constexpr int FAM = (BN == 128 ? 0 : BN == 64 ? 1 : 2) * 8;
        for (int q = 0; q < 6; ++q) atomicAdd(&si_tg_stamps[FAM + q], st_acc[q]);
        atomicAdd(&si_tg_stamps[FAM + 6], st_end - st_begin);
        atomicAdd(&si_tg_stamps[FAM + 7], 1ull);
    }
#endif
}

template <int MATH, int BM, int BN, int WARPS_M, int WARPS_N, int BK>
static int launch_cfg(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    typedef typename LdsElem<MATH>::type elem_t;
    constexpr int LD = BK + LdsElem<MATH>::PAD;
    constexpr int PLANES = (MATH == SI_MATH_BF16X3) ? 2 : 1;
    constexpr int NT = 64 * WARPS_M * WARPS_N;
    constexpr int MAXA = MaxA<BM, NT, BK>::value;
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int rowsA = (BM - 1) * p.stride + (p.ntaps - 1) * adil + 1;
    const int cap = (p.ntaps == 1 ? MAXA / 2 : MAXA) * NT;
    if (rowsA * (BK / 4) > cap)
        return si_fail(ctx, SI_EINVAL, "tapgemm: activation tile of %d rows exceeds the prefetch registers (stride %d, taps %d, dil %d)",
                       rowsA, p.stride, p.ntaps, p.dil);
    const int abufs = p.ntaps == 1 ? 2 : 1;
    size_t lds = (size_t)PLANES * ((size_t)abufs * rowsA + 2 * BN) * LD * sizeof(elem_t);
    if (lds > 160 * 1024) return si_fail(ctx, SI_EINVAL, "tapgemm: LDS tile of %zu bytes exceeds 160 KiB", lds);
    const bool lin = p.ntaps == 1;
    const bool a16 = p.x16 != nullptr;
    // row-contiguous epilogue of the 4-waves-per-SIMD kernels: needs columns in aligned groups of four and a 4.5 KB LDS
    // patch per wave (the operand tiles are dead by then)
    constexpr bool light_cfg = WavesPerSimd<BM, BN, NT, false, true>::value == 4;
    const bool wide = light_cfg && a16 && !lin && p.N % 4 == 0 && p.ldo % 4 == 0 && p.ooff % 4 == 0;
    if (wide) lds = std::max(lds, (size_t)(NT / 64) * 32 * 36 * sizeof(float));
    void (*kern)(const TapGemmParams) = lin ? tapgemm_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK, true> : tapgemm_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK, false>;
    if constexpr (MATH == SI_MATH_BF16 || MATH == SI_MATH_F16) {
        if (a16) kern = lin ? tapgemm_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK, true, true> : tapgemm_kernel<MATH, BM, BN, WARPS_M, WARPS_N, BK, false, true>;
    } else if (a16) {
        return si_fail(ctx, SI_EINVAL, "tapgemm: operand-ready (16-bit) activations need the bf16 or fp16 math mode");
    }
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    const int mtiles = (p.M + BM - 1) / BM;
    dim3 grid((unsigned)(p.nseg * mtiles * ((p.N + BN - 1) / BN)), (unsigned)p.groups);
    static const char* const math_names[] = {"f32", "bf16", "bf16x3", "f16"};
    char name[48];
    snprintf(name, sizeof(name), "tapgemm_%s_%dx%d%s", math_names[MATH], BM, BN, NT == 512 ? "w8" : "");
    const double macs = p.algo_macs > 0 ? p.algo_macs : (double)p.nseg * p.M * p.N * p.groups * (double)p.Cin * p.ntaps;
    // algorithmic HBM bytes: input once, every output copy once, residual / accumulate reads, weights once
    const double outs = (double)p.M * p.N * p.groups;
    double bytes = p.nseg * ((a16 ? 2.0 : 4.0) * p.Lin * p.Cin * p.groups + outs * ((p.out ? 4 : 0) + (p.out16 ? 2 : 0) + (p.res ? 4 : 0) + (p.res16 ? 2 : 0) + (p.accumulate ? (p.acc16 ? 2 : 4) : 0))) +
                   (double)p.groups * p.ntaps * p.N * p.Cin * (MATH == SI_MATH_F32 || MATH == SI_MATH_BF16X3 ? 4 : 2);
    TapGemmParams pk = p;
    pk.wide_epilogue = wide;
    if ((p.res16 || p.acc16) && MATH != SI_MATH_BF16 && MATH != SI_MATH_F16)
        return si_fail(ctx, SI_EINVAL, "tapgemm: 16-bit residual / accumulate exist in the bf16 and fp16 math modes only");
    si_prof_begin(ctx, si_prof_shape_name(name, p.M * (long)p.nseg, p.N * p.groups, p.Cin * p.ntaps), 2.0 * macs, bytes, st);   // (per shape under SI_PROF_SHAPES=1)
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, pk);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

template <int MATH, int BK>
static int launch_math(si_ctx* ctx, const TapGemmParams& p, hipStream_t st) {
    const int bn = si_pick_bn(p.N);
    if (bn == 128) {
        // One 8-wave workgroup per CU on a 256x128 tile (each weight slab staged once for twice the MFMAs) instead of two
        // 4-wave workgroups on 128x128 tiles: +6 % on the bf16x3 convolutions, +1 % in fp32.
        const int adil8 = p.dil < 0 ? -p.dil : p.dil;
        const int cap8 = (p.ntaps == 1 ? MaxA<256, 512>::value / 2 : MaxA<256, 512>::value) * 512;
        if constexpr (MATH == SI_MATH_F16 || MATH == SI_MATH_BF16) {
            // operand-ready Linear layers (the encoder's GEMMs): eight light waves (32 x 64 each, 109 VGPRs) per 128x128
            // tile instead of four 64x64 ones (211 VGPRs) -- twice the resident waves per CU; 2.83 vs 3.12 ms/step
            if (p.x16 && p.ntaps == 1 && BK == 32 && p.M > 256) {
                // 64-deep K chunks when K allows: half the iterations (barriers, waits) per tile
                if (p.Cin % 64 == 0) return launch_cfg<MATH, 128, 128, 4, 2, 64>(ctx, p, st);
                return launch_cfg<MATH, 128, 128, 4, 2, BK>(ctx, p, st);
            }
        }
        if (BK == 32 && p.M > 256 && (255 * p.stride + (p.ntaps - 1) * adil8 + 1) * (BK / 4) <= cap8)
            return launch_cfg<MATH, 256, 128, 4, 2, BK>(ctx, p, st);
        if constexpr (MATH == SI_MATH_F16 || MATH == SI_MATH_BF16) {
            // operand-ready convolutions whose halo rules out the 256-row tile (the encoder's stride-2 convs): light waves too
            if (p.x16 && p.ntaps > 1 && BK == 32 && p.M > 256 &&
                (127 * p.stride + (p.ntaps - 1) * adil8 + 1) * (BK / 4) <= MaxA<128, 512>::value * 512)
                return launch_cfg<MATH, 128, 128, 4, 2, BK>(ctx, p, st);
        }
        return launch_cfg<MATH, 128, 128, 2, 2, BK>(ctx, p, st);
    }
    // narrow N: 256-row tiles unless their halo'd activation tile would not fit the prefetch registers
    // (strided convs, Linear layers with their two half-sets), or the segment is so short that a 256-row tile
    // would be mostly padding
    const int adil = p.dil < 0 ? -p.dil : p.dil;
    const int cap = (p.ntaps == 1 ? MaxA<256>::value / 2 : MaxA<256>::value) * 256;
    const bool tall = ((255 * p.stride + (p.ntaps - 1) * adil + 1) * (BK / 4) <= cap) && p.M > 128;
    if constexpr (MATH == SI_MATH_F16 || MATH == SI_MATH_BF16) {
        // operand-ready activations, N = 64: eight light waves (32 rows x 64 columns, 92 VGPRs) per 256-row tile instead
        // of four heavy ones that spill at the 4-waves-per-SIMD budget (4.32 -> 4.20 ms/step; for N = 32 the 4-wave tile
        // stays faster: 3.11 vs 3.53)
        const bool fits8 = (255 * p.stride + (p.ntaps - 1) * adil + 1) * (BK / 4) <= MaxA<256, 512>::value * 512;
        if (bn == 64 && p.x16 && p.ntaps > 1 && tall && fits8 && BK == 32) return launch_cfg<MATH, 256, 64, 8, 1, BK>(ctx, p, st);
    }
    if (bn == 64) return tall ? launch_cfg<MATH, 256, 64, 4, 1, BK>(ctx, p, st) : launch_cfg<MATH, 128, 64, 2, 2, BK>(ctx, p, st);
    return tall ? launch_cfg<MATH, 256, 32, 4, 1, BK>(ctx, p, st) : launch_cfg<MATH, 128, 32, 4, 1, BK>(ctx, p, st);
}

#ifdef TG_STAMPS
extern "C" int si_debug_stamps(unsigned long long* out24, int reset) {
    if (out24 && hipMemcpyFromSymbol(out24, HIP_SYMBOL(si_tg_stamps), sizeof(si_tg_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(si_tg_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int si_launch_tapgemm(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st) {
    if (p.Cin % 16 != 0 || p.ldx % 4 != 0)
        return si_fail(ctx, SI_EINVAL, "tapgemm: Cin=%d must be a multiple of 16 and ldx=%d of 4", p.Cin, p.ldx);
    if (p.Npad % si_pick_bn(p.N) != 0 || p.Npad < p.N)
        return si_fail(ctx, SI_EINVAL, "tapgemm: Npad=%d does not match N=%d", p.Npad, p.N);
    if (p.M <= 0 || p.nseg <= 0) return SI_OK;
    if (!p.x == !p.x16) return si_fail(ctx, SI_EINVAL, "tapgemm: exactly one of x (fp32) and x16 (operand-ready) must be given");
    if (!p.out && !p.out16) return si_fail(ctx, SI_EINVAL, "tapgemm: no output");
    if (p.accumulate && !(p.acc16 ? (void*)p.out16 : (void*)p.out)) return si_fail(ctx, SI_EINVAL, "tapgemm: accumulate needs its output buffer");
    if (p.res && p.res16) return si_fail(ctx, SI_EINVAL, "tapgemm: give res or res16, not both");
    if (p.out16 && math != SI_MATH_BF16 && math != SI_MATH_F16)
        return si_fail(ctx, SI_EINVAL, "tapgemm: the 16-bit output copy exists in the bf16 and fp16 math modes only");
    // The epilogue masks through a buffer descriptor of olimit*4 bytes: byte offsets of every tile row (valid or not)
    // must stay below 2^31, and rows >= M must fall outside the descriptor.
    const long ooff_abs = p.ooff < 0 ? -p.ooff : p.ooff;
    if (((long)p.M + 256) * p.ldo * 4 + ooff_abs * 4 >= (1L << 31) || p.olimit * 4 >= (1L << 31) || (long)p.Lin * p.ldx >= (1L << 31))
        return si_fail(ctx, SI_EINVAL, "tapgemm: a segment of %ld floats exceeds the 32-bit in-segment byte offsets", p.olimit);
    if ((long)p.M * p.ldo + p.ooff < p.olimit)
        return si_fail(ctx, SI_EINVAL, "tapgemm: olimit=%ld must not exceed M*ldo+ooff=%ld (rows >= M are masked by the range check)",
                       p.olimit, (long)p.M * p.ldo + p.ooff);
    // the encoder's bf16 GEMMs and strided convolutions on operand-ready activations: the dedicated kernel (lingemm.hip)
    if (p.lingemm && math == SI_MATH_BF16 && p.x16 && p.groups == 1 && p.pad == 0 && p.dil == 1 && p.pro_slope == 1.f && p.alpha == 1.f &&
        !p.accumulate && !p.res16 && !p.acc16 && p.out16_slope == 1.f && p.ooff == 0 && (p.ntaps == 1 || p.ldx == p.Cin) && p.Npad == p.N &&
        p.olimit == (long)p.M * p.ldo && !p.seg_row_off && (!p.seg_m || (p.seg_orows == p.seg_m && p.olim_mul == p.ldo && p.seg_m_host))) {
        LinGemmParams q{};
        q.x16 = p.x16;
        const long xelems = p.nseg > 1 ? (long)p.nseg * p.x_seg_stride : (long)p.Lin * p.ldx;
        q.x_bytes = (int)(xelems * 2 < (1L << 31) ? xelems * 2 : 0);
        q.lda = p.stride * p.ldx; q.x_seg_stride = p.nseg > 1 ? p.x_seg_stride : 0;
        q.nseg = p.nseg; q.M = p.M; q.K = p.ntaps * p.Cin;
        q.w = static_cast<const unsigned short*>(p.w); q.w_bytes = p.ntaps * p.Npad * p.Cin * 2;
        q.N = p.N; q.Cin = p.Cin; q.ntaps = p.ntaps; q.w_tap_stride = (long)p.Npad * p.Cin;
        q.bias = p.bias; q.res = p.res; q.res_stats = p.res_stats; q.res_gamma = p.res_gamma; q.res_beta = p.res_beta; q.out = p.out; q.out16 = p.out16; q.ldo = p.ldo; q.o_seg_stride = p.nseg > 1 ? p.o_seg_stride : 0;
        q.act = p.act;
        q.seg_m = p.seg_m; q.seg_m_host = p.seg_m_host;                 // (ragged: rows >= seg_m[s] of a segment are neither computed nor stored)
        const int rc = si_launch_lingemm(ctx, q, st);
        if (rc <= 0) return rc;
    }
    if (p.res_stats) return si_fail(ctx, SI_EINVAL, "tapgemm: a LayerNorm residual (res_stats) is taken by the bf16 GEMM kernels only, which do not cover this shape");
    const bool k32 = (p.Cin % 32 == 0);
    switch (math) {
        case SI_MATH_F32: return k32 ? launch_math<SI_MATH_F32, 32>(ctx, p, st) : launch_math<SI_MATH_F32, 16>(ctx, p, st);
        case SI_MATH_BF16: return k32 ? launch_math<SI_MATH_BF16, 32>(ctx, p, st) : launch_math<SI_MATH_BF16, 16>(ctx, p, st);
        case SI_MATH_BF16X3: return k32 ? launch_math<SI_MATH_BF16X3, 32>(ctx, p, st) : launch_math<SI_MATH_BF16X3, 16>(ctx, p, st);
        case SI_MATH_F16: return k32 ? launch_math<SI_MATH_F16, 32>(ctx, p, st) : launch_math<SI_MATH_F16, 16>(ctx, p, st);
    }
    return si_fail(ctx, SI_EINVAL, "tapgemm: unknown math mode %d", math);
}
