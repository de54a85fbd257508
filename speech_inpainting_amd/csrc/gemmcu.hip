// gemmcu.hip -- the encoder's bf16 GEMM as ONE tile per compute unit (gfx950, wave64, v_mfma_f32_16x16x32_bf16, LDS-DMA staging).
//
//     out[seg][m][n] = epi( sum_k A[seg][m][k] * W[n][k] + bias[n] )        (the contract of lingemm.hip: LinGemmParams)
//
// Why a third kernel.  The transformer's GEMMs at the bench shape have M = B * T = 6368 rows (modeling_hubert.py:347-404: q|k|v,
// out-proj, FFN1, FFN2): a launch is ONE round of the chip whatever the tile, and what a CU then pays is the operand bytes its
// tiles draw through its L1 -- lingemm's two resident 96 x 128 tiles per CU draw 2 x (96 + 128) rows x K x 2 bytes: FFN2 (K = 3072)
// 2.75 MB per CU in 54 us = 50 GB/s, the rate of the vector-memory path measured in profiles/r03_lingemm_pmc.txt.  The bytes per CU
// are smallest when a CU's whole share of the output is ONE near-square tile: M * N / 256 outputs per CU is 276 x 276 for FFN1 and
// 138 x 138 for FFN2 / out-proj.  256 x 256 tiles (gemm256.hip) leave those shapes at 300 (1.17 rounds) and 75 tiles; this kernel
// takes the tile shape as a template parameter and the launcher picks, per shape, among the instantiations whose tiles fill whole
// rounds of the CUs, the one that draws the fewest bytes: 320 x 256 (FFN1: 20 x 12 = 240 tiles; conv1 as per-clip segments: 1280 =
// 5.0 rounds), 256 x 256 (QKV: 225), 160 x 128 (N = 768: 40 x 6 = 240), 208 x 256 / 224 x 128 / 128 x 128 (conv3-6, HuBERT-large).
//
// Structure.  NW waves (16 for the 320- and 256-row tiles, 8 for the others) as WM (M) x WN (N); a wave owns (16 MT) x (16 NT)
// outputs; one workgroup per CU.  Four waves per SIMD (<= 128 registers each) let the hardware interleave one wave's fragment reads
// with its neighbours' MFMAs -- no hand-made partner schedule as in gemm256.hip, whose 8 waves hold a whole K-tile's fragments.
// K-tiles of 64 (whole 128-byte lines of every operand row) arrive by LDS-DMA (buffer_load_dwordx4 ... lds, the XOR swizzle of
// lingemm.hip applied on the source side) into a ring of NS stages of (BM + BN) x 128 bytes; iteration t = [counted vmcnt: own
// pieces of K-tile t landed | barrier: everyone's landed, everyone done reading K-tile t - 1 | request K-tile t + NS - 1 into the
// stage K-tile t - 1 occupied | 2 k-steps of MT x NT MFMAs, one A fragment at a time with the next one's read issued under the
// current one's MFMAs].  One barrier per K-tile.  Epilogue from the accumulators (bias, fast erf-GELU, residual; bf16 through the
// half trade of gemm256.hip).  What bounds it -- the CU's own load path at ~47 GB/s, whatever the tile, wave count or ring depth --
// and everything that was tried against that: DESIGN.md 4.1e, profiles/r04_gemmcu_ab.txt.
//
// BIT-IDENTICAL to lingemm.hip and gemm256.hip: every output's sum runs over K in steps of 32 through the same MFMA with the same
// operand roles (D^T = W A^T) and the same epilogue arithmetic, so the launcher may choose by shape and batch size without a clip's
// result depending on it (tests/test_gpu_respair.py::test_gemmcu_*).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int C_BK = 64, C_ROWB = C_BK * 2;                // 128-byte LDS rows

#define C_LDS(off) ((__attribute__((address_space(3))) void*)(smem + (off)))

__device__ __forceinline__ void c_wait_vm(int n) {                      // n is wave-uniform: scalar branches
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;    // (never wrong: waits for more)
    }
}

// Diagnostic build only (-DGCU_TIMELINE; tools/exp_encoder_only.py): wall-clock totals of every workgroup's wave 0 in 100 MHz ticks,
// [class: 0 = N 3072, 1 = N 2304, 2 = N 768 with K 3072, 3 = other][0 prologue (to the first K-tile visible), 1 K loop, 2 epilogue, 3 workgroups]
#ifdef GCU_TIMELINE
__device__ unsigned long long gcu_tl[4][4];
#define C_TL_DECL unsigned long long tl_t = wall_clock64(), tl_acc[3] = {0, 0, 0};
#define C_TL(ph) { const unsigned long long n_ = wall_clock64(); tl_acc[ph] += n_ - tl_t; tl_t = n_; }
#define C_TL_FLUSH if (threadIdx.x == 0) { const int c_ = p.N == 3072 ? 0 : (p.N == 2304 ? 1 : ((p.N == 768 && p.K == 3072) ? 2 : 3)); for (int q_ = 0; q_ < 3; ++q_) atomicAdd(&gcu_tl[c_][q_], tl_acc[q_]); atomicAdd(&gcu_tl[c_][3], 1ull); }
extern "C" int si_debug_gcu_timeline(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gcu_tl), sizeof(gcu_tl)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4][4] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(gcu_tl), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define C_TL_DECL
#define C_TL(ph)
#define C_TL_FLUSH
#endif

// TC ("transposed convolution" mode, the generator's early upsamplers on the fp16 stream -- I_ea/hifi_gan/models.py:87-95,110-111:
// x = lrelu(x, 0.1); x = ups[i](x)): fp16 MFMA and fp16 output (one saturating rounding: MODE.FP16_OVFL); tap t of GEMM row m reads
// input row m + t * tc_dil of the SEGMENT (p.tc_dil = -1: ConvTranspose1d with k = 2 stride as two taps, api.hip) through a
// descriptor over the segment's own rows, whose range check returns the zeros of the convolution's padding (row -1, row Lin); the
// leaky-ReLU of the input is applied to the A fragments (max(x, slope x) on packed halves: tapgemm.hip's select for 0 < slope < 1);
// the output row is shifted by p.tc_ooff elements and cropped to [0, p.tc_olimit) of its segment.
template <int NW, int WM, int WN, int MT, int NT, int NS, bool TC = false>
__global__ __launch_bounds__(NW * 64) void gemmcu_kernel(const LinGemmParams p) {
    static_assert(WM * WN == NW, "waves");
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int STAGE = (BM + BN) * C_ROWB;
    constexpr int PT = (BM + BN) / 8, PA = BM / 8;                      // 1-KB DMA pieces (8 rows) per K-tile: A rows first, then W rows
    constexpr int PW = (PT + NW - 1) / NW;
    static_assert(NS * STAGE <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [NS][A BM rows | W BN rows][128 bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int r16 = lane & 15, kg = lane >> 4;

    // tile walk: row-block-major order cut into 8 contiguous runs, one per XCD (workgroup ids b and b + 8 share an L2), as gemm256.hip
    const int ntn = p.N / BN;
    const int mtiles = (p.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int run0 = (int)((long)p.xcd_rows * xcd / 8), run1 = (int)((long)p.xcd_rows * (xcd + 1) / 8);
    const int tile = run0 + slot;
    if (tile >= run1) return;
    const int mtx = tile / ntn;
    int seg, m0, Ms = p.M;
    if (p.seg_m) {                                                     // ragged batches: row blocks numbered segment by segment without gaps
        const SiVlTile v = si_vl_tile(p.seg_m, p.nseg, BM, mtx);
        seg = v.b; m0 = v.row0; Ms = v.L;
    } else {
        seg = mtx / mtiles;
        m0 = (mtx - seg * mtiles) * BM;
    }
    const int n0 = (tile - mtx * ntn) * BN;

    if constexpr (TC) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);   // MODE.FP16_OVFL: conversions to fp16 saturate at +-65504
    const int tc_rows = (TC && p.tc_seg_lin) ? p.tc_seg_lin[seg] : p.tc_rows_in;          // (ragged batches: the clip's own rows; scalar loads)
    const long tc_olimit = (TC && p.tc_seg_orows) ? (long)p.tc_seg_orows[seg] * p.tc_olim_mul : p.tc_olimit;
    const __amdgpu_buffer_rsrc_t arsrc = TC ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x16) + (long)seg * p.x_seg_stride, 0, tc_rows * p.lda * 2, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x16), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.w_bytes, 0x00020000);
    // LDS-DMA: a piece = 8 rows; lane l lands at physical chunk l & 7 of row l >> 3 and fetches the logical chunk the swizzle puts there
    const int srow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((srow >> 1) & 3) << 1);
    // (TC: the segment offset is in the descriptor and EVERY row term in the vector offset -- the range check that supplies the zero
    //  rows looks at the vector offset; a negative one is a huge unsigned one)
    const int a_lane = TC ? (m0 + srow) * p.lda * 2 + lchunk * 16 : (int)(((long)seg * p.x_seg_stride + (long)(m0 + srow) * p.lda) * 2) + lchunk * 16;
    const int w_lane = (int)(((long)(n0 + srow) * p.Cin) * 2) + lchunk * 16;
    const int cpt = p.Cin / C_BK;
    const int nk = p.K / C_BK;
    const int mine = (PT - wave + NW - 1) / NW;                             // pieces this wave requests per K-tile (wave-uniform)
    auto stage = [&](int kt) {
        const int buf = (kt % NS) * STAGE;
        const int tap = kt / cpt;
        const long koff = ((long)tap * p.w_tap_stride + (long)(kt - tap * cpt) * C_BK) * 2;
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int piece = wave + NW * q;
            if (piece >= PT) break;
            if (piece < PA) {
                if constexpr (TC) {
                    const int rowterm = __builtin_amdgcn_readfirstlane((piece * 8 + tap * p.tc_dil) * p.lda * 2);
                    const int soff = __builtin_amdgcn_readfirstlane((kt - tap * cpt) * C_BK * 2);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, C_LDS(buf + piece * 1024), 16, a_lane + rowterm, soff, 0, 0);
                } else {
                    const int soff = __builtin_amdgcn_readfirstlane(kt * C_BK * 2 + piece * 8 * p.lda * 2);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, C_LDS(buf + piece * 1024), 16, a_lane, soff, 0, 0);
                }
            } else {
                const int soff = __builtin_amdgcn_readfirstlane((int)(koff + (long)(piece - PA) * 8 * p.Cin * 2));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, C_LDS(buf + piece * 1024), 16, w_lane, soff, 0, 0);
            }
        }
    };

    // fragment reads: lane (r16, kg) reads chunk (4 ks + kg) ^ swizzle of row (16 * tile + r16)
    const int foff = r16 * C_ROWB + ((kg << 4) ^ ((((r16 >> 1) & 3) << 1) << 4));
    const int a_off = wr * MT * 16 * C_ROWB + foff;
    const int w_off = (BM + wc * NT * 16) * C_ROWB + foff;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    C_TL_DECL
    const int pre = NS - 1 < nk ? NS - 1 : nk;                          // K-tiles requested ahead
    for (int kt = 0; kt < pre; ++kt) stage(kt);
    for (int t = 0; t < nk; ++t) {
        // K-tiles younger than t that this wave has in flight: min(NS - 2, nk - 1 - t)
        const int younger = (nk - 1 - t) < (NS - 2) ? (nk - 1 - t) : (NS - 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else c_wait_vm(younger * mine);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (every fragment read of K-tile t - 1 retired before its stage is re-requested)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef GCU_TIMELINE
        if (t == 0) C_TL(0)
#endif
        if (t + NS - 1 < nk) stage(t + NS - 1);
        const char* buf = smem + (t % NS) * STAGE;
        {
            // One A fragment at a time, two registers sets deep: the fragment of step s + 1 (s = ks * MT + i) is requested before the NT
            // MFMAs of step s, and the W fragments of the second k-step take the registers of the first as its last MFMAs release
            // them -- every read but the K-tile's first NT + 1 is issued under MFMAs of the same wave.  Each accumulator still sums
            // k-step 0 before k-step 1: the values do not change.
            bf16x8 fw[NT], fa[2];
            const _Float16 slope = (_Float16)p.tc_slope;
            const bool has_act = TC && p.tc_slope != 1.f;              // (uniform: an input its producer already activated skips the VALU work)
            auto act = [&](bf16x8 v) {                                 // TC: leaky-ReLU on the packed halves of an A fragment
                if constexpr (TC) {
                    if (!has_act) return v;
                    const f16x8 h = __builtin_bit_cast(f16x8, v);
                    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(h, h * slope));
                } else {
                    return v;
                }
            };
            auto mma = [&](bf16x8 w, bf16x8 a, f32x4 c) {
                if constexpr (TC) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
                else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
            };
#pragma unroll
            for (int j = 0; j < NT; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(buf + (w_off + j * 16 * C_ROWB));
            fa[0] = act(*reinterpret_cast<const bf16x8*>(buf + a_off));
#pragma unroll
            for (int s2 = 0; s2 < 2 * MT; ++s2) {
                const int ks = s2 / MT, i = s2 - ks * MT;
                if (s2 + 1 < 2 * MT) {
                    const int ks1 = (s2 + 1) / MT, i1 = (s2 + 1) - ks1 * MT;
                    fa[(s2 + 1) & 1] = act(*reinterpret_cast<const bf16x8*>(buf + ((a_off ^ (ks1 * 64)) + i1 * 16 * C_ROWB)));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = mma(fw[j], fa[s2 & 1], acc[i][j]);
                    if (s2 == MT - 1) fw[j] = *reinterpret_cast<const bf16x8*>(buf + ((w_off ^ 64) + j * 16 * C_ROWB));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    C_TL(1)
    // ---- epilogue from the accumulators: lane (r16, kg) holds row 16 i + r16, columns 16 j + 4 kg + [0, 4) of the wave's tile
    const bool gelu = p.act == SI_ACT_GELU;
    const bool has_res = p.res != nullptr;
    const bool res_ln = p.res_stats != nullptr;
    const long obase = (long)seg * p.o_seg_stride;
    const int ncol0 = n0 + wc * NT * 16;
    f32x4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + 16 * j + 4 * kg);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wr * MT * 16 + 16 * i + r16;
        const bool live = m < Ms;
        const long orow = obase + (long)(live ? m : Ms - 1) * p.ldo + ncol0;      // dead rows read row M - 1 and store nothing
        f32x4 rv[NT];
        if (has_res) {
#pragma unroll
            for (int j = 0; j < NT; ++j) rv[j] = *reinterpret_cast<const f32x4*>(p.res + orow + 16 * j + 4 * kg);
            if (res_ln) {                                              // the residual is LayerNorm(res): si_ln_apply on the row's (mean, rstd)
                const f32x2 ms = *reinterpret_cast<const f32x2*>(p.res_stats + 2 * (long)(live ? m : Ms - 1));
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const f32x4 lg = *reinterpret_cast<const f32x4*>(p.res_gamma + ncol0 + 16 * j + 4 * kg);
                    const f32x4 lb = *reinterpret_cast<const f32x4*>(p.res_beta + ncol0 + 16 * j + 4 * kg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[j][e] = si_ln_apply(rv[j][e], ms[0], ms[1], lg[e], lb[e]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 v = acc[i][j] + bv[j];
            if (gelu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = si_gelu_fast(v[e]);
            }
            if (has_res) v += rv[j];
            if (p.out && live) *reinterpret_cast<f32x4*>(p.out + orow + 16 * j + 4 * kg) = v;
            acc[i][j] = v;
        }
        if (p.out16) {
            if constexpr (NT % 2 == 0) {
                // lanes l and l + 16 trade halves of a column-tile pair (v_permlane16_swap): 16 bytes per lane, 64 contiguous per row
#pragma unroll
                for (int t = 0; t < NT / 2; ++t) {
                    u32x2 p0, p1;
                    if constexpr (TC) {
                        p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[i][2 * t], f16x4));
                        p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[i][2 * t + 1], f16x4));
                    } else {
                        p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[i][2 * t], bf16x4));
                        p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[i][2 * t + 1], bf16x4));
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const auto r = __builtin_amdgcn_permlane16_swap(p0[q], p1[q], false, false);
                        p0[q] = r[0]; p1[q] = r[1];
                    }
                    const int col = 16 * (2 * t + (kg & 1)) + 4 * (kg & ~1);
                    if constexpr (TC) {
                        // the shifted, cropped output row: element (m, n) lives at m * ldo + n + tc_ooff of its segment; a chunk of 8 is
                        // wholly inside or outside [0, tc_olimit) (ooff, olimit and the channel count are multiples of 8)
                        const long flat = (long)m * p.ldo + ncol0 + col + p.tc_ooff;
                        if (live && flat >= 0 && flat + 8 <= tc_olimit) *reinterpret_cast<u32x4*>(p.out16 + obase + flat) = u32x4{p0[0], p0[1], p1[0], p1[1]};
                    } else {
                        if (live) *reinterpret_cast<u32x4*>(p.out16 + orow + col) = u32x4{p0[0], p0[1], p1[0], p1[1]};
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (live) *reinterpret_cast<bf16x4*>(p.out16 + orow + 16 * j + 4 * kg) = __builtin_convertvector(acc[i][j], bf16x4);
            }
        }
    }
#ifdef GCU_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    C_TL(2)
    C_TL_FLUSH
}

struct CuCfg { int bm, bn, ns; };
static const CuCfg k_cfgs[] = {{320, 256, 2}, {256, 256, 2}, {160, 128, 4}, {224, 128, 3}, {128, 128, 4}, {208, 256, 2}};
constexpr int k_ncfg = 6, k_nrule = 6;

template <int NW, int WM, int WN, int MT, int NT, int NS, bool TC = false>
static int gemmcu_launch(si_ctx* ctx, const LinGemmParams& p, hipStream_t st, double algo_macs = 0.0) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    const size_t lds = (size_t)NS * (BM + BN) * C_ROWB;
    auto kern = gemmcu_kernel<NW, WM, WN, MT, NT, NS, TC>;
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    LinGemmParams q = p;
    const int mtiles = (p.M + BM - 1) / BM;
    const long row_blocks = p.seg_m_host ? si_vl_tiles(p.seg_m_host, p.nseg, BM) : (long)p.nseg * mtiles;   // ragged batches: no gaps
    const int tiles = (int)(row_blocks * (p.N / BN));
    q.xcd_rows = tiles;
    const unsigned grid = (unsigned)((tiles + 7) / 8 * 8);
    double rows_real = (double)p.nseg * p.M;
    if (p.seg_m_host) { rows_real = 0; for (int sg = 0; sg < p.nseg; ++sg) rows_real += p.seg_m_host[sg]; }
    const double macs = algo_macs > 0.0 ? algo_macs : rows_real * p.N * (double)p.K;
    const double outs = rows_real * p.N;
    const double bytes = 2.0 * (rows_real * p.lda + p.nseg * (double)(p.K - p.lda > 0 ? p.K - p.lda : 0)) + outs * ((p.out ? 4 : 0) + (p.out16 ? 2 : 0) + (p.res ? 4 : 0)) + 2.0 * p.N * p.K;
    char name[48];
    snprintf(name, sizeof(name), TC ? "gemmcu_f16_%dx%d" : "gemmcu_bf16_%dx%d", BM, BN);        // one family per instantiation, as rocprofv3 lists them
    si_prof_begin(ctx, si_prof_shape_name(name, p.M * (long)p.nseg, p.N, p.K), 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, q);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered or the rule leaves it to the other kernels.
// SI_ENC_GEMMCU: 0 never; 1 (default) by the rule below; 2 every shape an instantiation covers (the bit-identity tests);
// 10 + c: instantiation c wherever it covers the shape (A/B runs).
int si_launch_gemmcu(si_ctx* ctx, const LinGemmParams& p, hipStream_t st) {
    const int opt = si_opt_gemmcu(ctx);
    if (opt == 0) return 1;
    if (p.seg_m && !p.seg_m_host) return 1;                            // ragged segments need their host copy for the grid
    if (p.res_stats && p.nseg != 1) return 1;                          // a LayerNorm residual: one flat segment (the row index is the stats index)
    if (p.Cin % C_BK || p.K % C_BK || p.K != p.ntaps * p.Cin || p.ldo % 4 || p.lda % 8 || p.M <= 0 || p.nseg <= 0 || p.N % 128) return 1;
    if (p.x_bytes <= 0 || p.w_bytes <= 0 || (long)p.nseg * p.x_seg_stride * 2 + (long)(p.M + 320) * p.lda * 2 >= (1L << 31)) return 1;
    if ((long)(p.N + 256) * p.Cin * 2 + (long)p.ntaps * p.w_tap_stride * 2 >= (1L << 31)) return 1;
    const int cus = si_num_cus(ctx);
    double rows_real = (double)p.nseg * p.M;
    if (p.seg_m_host) { rows_real = 0; for (int sg = 0; sg < p.nseg; ++sg) rows_real += p.seg_m_host[sg]; }
    int pick = -1;
    if (opt >= 10) {
        pick = opt - 10;
        if (pick >= k_ncfg || p.N % k_cfgs[pick].bn) return 1;
    } else {
        // Whole rounds of the chip: among the instantiations whose tiles fill their rounds -- tiles / (rounds x CUs) >= 0.6 for one
        // round (below that the 128-row kernels' two workgroups per CU spread the same bytes over more L1s: HuBERT-large's N = 1024
        // GEMMs at B = 16), >= 0.75 for several -- and whose rows are at least 3/4 real (per-clip segments of 199 rows on 160-row
        // tiles are not), the one with the fewest operand bytes per CU: rounds x (BM + BN) rows of K.  The feature extractor's
        // convolutions are several rounds (conv1: 1280 tiles of 320 x 256 = 5.0 rounds, against 6.25 of 256 x 256); the
        // transformer's GEMMs one.  opt == 2 drops the conditions.
        double best = 1e30;
        for (int c = 0; c < k_nrule; ++c) {
            if (p.N % k_cfgs[c].bn) continue;
            const long rb = p.seg_m_host ? si_vl_tiles(p.seg_m_host, p.nseg, k_cfgs[c].bm) : (long)p.nseg * ((p.M + k_cfgs[c].bm - 1) / k_cfgs[c].bm);
            if (rb <= 0) continue;
            const long tiles = rb * (p.N / k_cfgs[c].bn);
            const long rounds = (tiles + cus - 1) / cus;
            const double fill = (double)tiles / (double)(rounds * cus);
            if (opt == 1 && (fill < (rounds == 1 ? 0.6 : 0.75) || rows_real < 0.75 * (double)rb * k_cfgs[c].bm)) continue;
            const double cost = (double)rounds * (k_cfgs[c].bm + k_cfgs[c].bn);
            if (cost < best) { best = cost; pick = c; }
        }
        if (pick < 0) return 1;
    }
    switch (pick) {
        case 0: return gemmcu_launch<16, 4, 4, 5, 4, 2>(ctx, p, st);     // 320 x 256, 16 waves of 80 x 64
        case 1: return gemmcu_launch<16, 4, 4, 4, 4, 2>(ctx, p, st);     // 256 x 256, 16 waves of 64 x 64
        case 2: return gemmcu_launch<8, 2, 4, 5, 2, 4>(ctx, p, st);      // 160 x 128, 8 waves of 80 x 32, four-stage ring
        case 3: return gemmcu_launch<8, 2, 4, 7, 2, 3>(ctx, p, st);      // 224 x 128, 8 waves of 112 x 32 (HuBERT-large's N = 1024 GEMMs at M = 6368)
        case 4: return gemmcu_launch<8, 2, 4, 4, 2, 4>(ctx, p, st);      // 128 x 128, 8 waves of 64 x 32 (the same at M = 3184)
        default: return gemmcu_launch<8, 1, 8, 13, 2, 2>(ctx, p, st);    // 208 x 256, 8 waves of 208 x 32 (HuBERT-large's FFN1: N = 4096 at M = 6368 / 3184)
    }
}

// The generator's early upsamplers on the fp16 stream (I_ea/hifi_gan/models.py:87-95,110-111; api.hip hands a ConvTranspose1d with
// k = 2 stride as a two-tap convolution with dil = -1 and N = stride * Cout): the TC instantiations of the kernel above.
// the instantiation si_launch_gemmcu_tc would use for this layer (0: 256 x 256, 1: 192 x 256), or -1 when it leaves it to the tap-GEMM
static int gemmcu_tc_pick(si_ctx* ctx, const TapGemmParams& p, bool always) {
    if (si_opt_gemmcu(ctx) == 0) return -1;
    if (!p.x16 || p.x || p.out || !p.out16 || p.out16_slope != 1.f || p.dil != -1 || p.ntaps != 2 || p.stride != 1 || p.pad != 0 || p.groups != 1) return -1;
    if (p.res || p.res16 || p.accumulate || p.acc16 || p.alpha != 1.f || p.act != SI_ACT_NONE || p.Npad != p.N || p.ldx != p.Cin) return -1;
    const bool vl = p.seg_lin || p.seg_m || p.seg_orows || p.seg_row_off;
    if (vl && (!p.seg_lin || !p.seg_m || !p.seg_orows || !p.seg_m_host || p.seg_row_off || p.olim_mul % 8)) return -1;   // ragged batches: per-clip rows in / rows / rows out
    if (p.Cin % C_BK || p.N % 256 || p.ldo % 8 || p.ooff % 8 || p.olimit % 8 || p.o_seg_stride % 8 || !(p.pro_slope > 0.f && p.pro_slope <= 1.f)) return -1;
    if (p.M <= 0 || p.nseg <= 0 || (long)(p.Lin + 1) * p.ldx * 2 >= (1L << 30) || (long)(p.M + 320) * p.ldx * 2 >= (1L << 30)) return -1;
    if ((long)(p.N + 256) * p.Cin * 2 + (long)p.ntaps * p.Npad * p.Cin * 2 >= (1L << 31)) return -1;
    // tile height by the kernel's cost rule over 256 / 192 rows (BN = 256): rounds x (BM + BN), rows at least 3/4 real (the 320-row
    // instantiation spills inside its K loop with the activation on the fragments: not built)
    const int cus = si_num_cus(ctx);
    int pick = -1;
    double best = 1e30;
    const int bms[2] = {256, 192};
    for (int c = 0; c < 2; ++c) {
        const long rb = vl ? si_vl_tiles(p.seg_m_host, p.nseg, bms[c]) : (long)p.nseg * ((p.M + bms[c] - 1) / bms[c]);
        if (rb <= 0) continue;
        const long tiles = rb * (p.N / 256);
        const long rounds = (tiles + cus - 1) / cus;
        double rows_real = (double)p.nseg * p.M;
        if (vl) { rows_real = 0; for (int sg = 0; sg < p.nseg; ++sg) rows_real += p.seg_m_host[sg]; }
        if (!always && rows_real < 0.75 * (double)rb * bms[c]) continue;                 // (always: the tests' one-frame clips)
        const double cost = (double)rounds * (bms[c] + 256);
        if (cost < best) { best = cost; pick = c; }
    }
    return pick;
}

bool si_gemmcu_tc_covers(si_ctx* ctx, const TapGemmParams& p, bool always) { return gemmcu_tc_pick(ctx, p, always) >= 0; }

int si_launch_gemmcu_tc(si_ctx* ctx, const TapGemmParams& p, hipStream_t st, bool always) {
    const int pick = gemmcu_tc_pick(ctx, p, always);
    if (pick < 0) return 1;
    LinGemmParams q{};
    q.x16 = p.x16; q.x_bytes = 0; q.lda = p.ldx; q.x_seg_stride = p.x_seg_stride;
    q.nseg = p.nseg; q.M = p.M; q.K = p.ntaps * p.Cin;
    q.w = static_cast<const unsigned short*>(p.w); q.w_bytes = p.ntaps * p.Npad * p.Cin * 2;
    q.N = p.N; q.Cin = p.Cin; q.ntaps = p.ntaps; q.w_tap_stride = (long)p.Npad * p.Cin;
    q.bias = p.bias; q.out16 = p.out16; q.ldo = p.ldo; q.o_seg_stride = p.o_seg_stride;
    q.act = SI_ACT_NONE;
    q.tc_dil = p.dil; q.tc_rows_in = p.Lin; q.tc_slope = p.pro_slope; q.tc_ooff = p.ooff; q.tc_olimit = p.olimit;
    q.seg_m = p.seg_m; q.seg_m_host = p.seg_m_host; q.tc_seg_lin = p.seg_lin; q.tc_seg_orows = p.seg_orows; q.tc_olim_mul = p.olim_mul;
    if (pick == 0) return gemmcu_launch<16, 4, 4, 4, 4, 2, true>(ctx, q, st, p.algo_macs);
    return gemmcu_launch<16, 4, 4, 3, 4, 2, true>(ctx, q, st, p.algo_macs);
}
