// lingemm.hip -- the encoder's bf16 GEMM (gfx950, wave64, v_mfma_f32_16x16x32_bf16).
//
//     out[seg][m][n] = epi( sum_k A[seg][m][k] * W[n][k] + bias[n] )        A row m = K consecutive bf16 at x16 + m * lda
//
// covers, on operand-ready bf16 activations (the bf16 encoder mode):
//   * every Linear of the transformer (modeling_hubert.py:262-368: q/k/v fused, out-proj + residual, FFN1 + GELU, FFN2 +
//     residual) and the feature projection (:216-231): lda = K;
//   * the six strided feature-extractor convolutions (:106-124; k = 3 / 2, stride 2, no padding): on channels-last
//     activations the k taps of output row m are the k * Cin CONSECUTIVE values starting at input row m * stride, so the
//     convolution is this GEMM with overlapping A rows (lda = stride * Cin, K = k * Cin) and the tap-major weight blocks
//     W[tap][n][ci] read as K chunks.
// It replaces the generic tap-GEMM (tapgemm.hip: 128x128 tile, eight 32x64 waves, 32x32x16 MFMA, 16-byte row padding)
// for these shapes: 64x64 wave tiles cut the LDS reads per MFMA by a third, the 16x16x32 MFMA shape holds a higher
// clock under load (tools/ubench/mfma_rate.hip), a 64-deep K chunk feeds 32 MFMAs per wave between barriers instead of
// 8, and the output goes through an LDS image so that every global access is a whole row segment.
//
// (Measured and rejected: a 256 x 128 tile with eight waves and one workgroup per CU for the large-M convolutions --
// a quarter fewer operand bytes out of L2 per flop -- ran conv1 / conv2 at 520 / 269 us against 481 / 248: what these
// GEMMs need is the second resident workgroup, not fewer bytes.  Operand tiles moved by LDS-DMA (buffer_load_dwordx4 ...
// lds, swizzle applied on the source side, no staging registers or ds_write: 144 VGPRs instead of 210) gave the same
// times within 1-2 % on every shape in a same-box A/B -- the VGPR -> LDS store path is not what limits this kernel.)
//
// Tile height per shape (same-box, us per launch at B = 32; BM = 128 / 96 / 64): conv1 472 / 521 / 514, conv4 73 / 79 /
// 69, conv6 24 / 24 / 21, QKV 41 / 46 / 44, out-proj 25 / 21 / 21, FFN1 66 / 71 / 64, FFN2 62 / 53 / 57 -- a shorter tile
// costs LDS bytes per MFMA but fills the workgroup slots when M = 6368 leaves 300 tall tiles for 512 slots; the
// launcher's rule (rounds x (BM + 24)) picks 128 / 128 / 64 / 128 / 96 / 64 / 96 for these: 194 -> 174 us per
// transformer layer.
//
// What bounds it (rocprofv3 counters, profiles/r03_lingemm_pmc.txt): the vector-memory path.  A 128 x 128 tile draws 32 KB of
// operands from L2 per 2.1 MFLOP; at the measured rate that is 45 GB/s per CU (66-73 have been seen at best), the texture
// addresser is stalled by the L1 45 % of the time, SQ_VMEM_TA_*_FIFO_FULL 18-20 % of the wave cycles, matrix pipe busy 0.29.
// Two experiments confirmed it: a form with the A operand fetched straight into registers in the MFMA layout (half the LDS
// staging, a three-deep W ring, the fragments of the next chunk requested before the barrier, the epilogue from the
// accumulators) and a 13-operation erf instead of libm's erff in the epilogue both left every shape's time unchanged
// within 2 %.  The lever is bytes per flop: gemm256.hip (256 x 256 tiles, 7.8 KB per MFLOP) takes the shapes whose tiles fill
// the chip; this kernel keeps the rest.  Both order every output's sum identically (K in steps of 32, the same MFMA with the
// same operand roles, the same epilogue arithmetic), so their results are BIT-IDENTICAL and the choice between them may
// depend on the batch size without a clip's result depending on it.
//
// Structure (as respair_wide.hip): one 4-wave workgroup per BM x 128 tile (BM = 128 below), two workgroups per CU.  K chunks of 64
// stream global -> registers -> LDS through a double buffer with the stores spread behind the MFMA blocks; one barrier
// per chunk.  Orientation D^T = W * A^T: a lane holds one output row (column l & 15) and four consecutive output
// columns, so the fp32 output image is written with 16-byte stores.  LDS rows are 128 bytes, chunk c of row r lives at
// chunk c ^ (((r >> 1) & 3) << 1): conflict-free for the 16x16x32 operand read (see respair_wide.hip).
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


constexpr int LG_BN = 128, LG_BK = 64, LG_NT = 256;
constexpr int LG_ROWB = LG_BK * 2;                                     // 128-byte LDS rows
constexpr int LG_WTILE = LG_BN * LG_ROWB;                              // 16 KB of weights per buffer

__device__ __forceinline__ int lg_swz(int row) { return ((row >> 1) & 3) << 5; }

// accumulators -> fp32 image of the tile in LDS -> bias / GELU / residual -> row-contiguous stores
template <int BM>
__device__ __forceinline__ void lg_epilogue(const LinGemmParams& p, char* smem, const f32x4 (&acc)[BM / 32][4], int tid, int wm0, int wn0, int r16, int kg,
                                            int seg, int m0, int n0, int Ms) {
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
        const int m = wm0 + 16 * i + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = ((wn0 + 16 * j) >> 2) + kg;                 // 16-byte chunk (4 columns) of the 512-byte image row
            *reinterpret_cast<f32x4*>(smem + m * 512 + ((co ^ (m & 15)) << 4)) = acc[i][j];
        }
    }
    __syncthreads();
    const int c4 = tid & 31, er0 = tid >> 5;                           // a lane owns 4 consecutive columns of rows er0 + 8 it
    const int n = n0 + 4 * c4;
    const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    const long obase = (long)seg * p.o_seg_stride;
    const bool gelu = p.act == SI_ACT_GELU;
    // res_stats: the residual is LayerNorm(res), recomputed from the rows before the normalisation (common.h: si_ln_apply)
    const bool res_ln = p.res_stats != nullptr;
    f32x4 lg = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if (res_ln) { lg = *reinterpret_cast<const f32x4*>(p.res_gamma + n); lb = *reinterpret_cast<const f32x4*>(p.res_beta + n); }
#pragma unroll 4
    for (int it = 0; it < BM / 8; ++it) {
        const int r = er0 + 8 * it;
        const int m = m0 + r;
        if (m >= Ms) break;                                            // rows ascend with `it`
        const long o = obase + (long)m * p.ldo + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * 512 + ((c4 ^ (r & 15)) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e] + bv[e];
            if (gelu) x = si_gelu_fast(x);
            v[e] = x;
        }
        if (p.res) {
            f32x4 rr = *reinterpret_cast<const f32x4*>(p.res + o);
            if (res_ln) {
                const f32x2 ms = *reinterpret_cast<const f32x2*>(p.res_stats + 2 * (long)m);
#pragma unroll
                for (int e = 0; e < 4; ++e) rr[e] = si_ln_apply(rr[e], ms[0], ms[1], lg[e], lb[e]);
            }
            v += rr;
        }
        if (p.out) *reinterpret_cast<f32x4*>(p.out + o) = v;
        if (p.out16) *reinterpret_cast<bf16x4*>(p.out16 + o) = __builtin_convertvector(v, bf16x4);
    }
}

// BM = rows of the tile (64 / 96 / 128): 2 x 2 waves of (BM / 2) x 64.  The launcher picks BM per shape so that the
// tiles fill the 2 x CUs workgroup slots in as few, as full rounds as possible (M = 6368 leaves 300 tiles of 128 rows
// for N = 768: most CUs then run ONE workgroup with nothing to overlap its loads and epilogue with).
template <int BM>
__global__ __launch_bounds__(LG_NT, 2) void lingemm_kernel(const LinGemmParams p) {
    constexpr int MT = BM / 32;                                        // 16-row MFMA tiles per wave
    constexpr int ATILE = BM * LG_ROWB;                                // bytes of the A tile per buffer
    constexpr int BUF = ATILE + LG_WTILE;
    constexpr int LG_BM = BM;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][A BM x 64 | W 128 x 64] bf16; later the fp32 output image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * 64;

    const int ntn = p.N / LG_BN;
    const int mtiles = (p.M + LG_BM - 1) / LG_BM;
    int tile = blockIdx.x;
    if (p.xcd_rows > 0) {
        // XCD-aware order: workgroup ids are dealt round-robin over the 8 XCDs, so ids b and b + 8 share an L2.  All N
        // tiles of one row block go to ONE XCD (its A rows are then fetched into one L2 instead of eight): row block
        // mt = xcd + 8 * (slot / ntn), column tile = slot % ntn; the grid is padded to 8 * ceil(row blocks / 8) row blocks.
        const int xcd = tile & 7, slot = tile >> 3;
        const int mtx = xcd + 8 * (slot / ntn);
        if (mtx >= p.xcd_rows) return;
        tile = mtx * ntn + slot % ntn;
    }
    const int mt = tile / ntn;                                         // N tiles of one M tile share A rows
    const int seg = mt / mtiles;
    const int m0 = (mt - seg * mtiles) * LG_BM;
    const int n0 = (tile - mt * ntn) * LG_BN;
    const int Ms = p.seg_m ? p.seg_m[seg] : p.M;                       // ragged batches: this segment's own rows (scalar load)
    if (m0 >= Ms) return;

    // A rows through a descriptor over the whole activation buffer (reads past its end return zero; rows >= M of the last
    // tile of a segment read the next segment's data or zero and only feed output rows that are never stored)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x16), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.w_bytes, 0x00020000);
    const int sc = tid & 7, sr0 = tid >> 3;                            // staging: 16-byte chunk sc of rows sr0 + 32 i
    const int a_voff = (int)(((long)seg * p.x_seg_stride + (long)(m0 + sr0) * p.lda) * 2) + sc * 16;
    const int a_step = p.lda * 2 * 32;
    const int w_voff = ((n0 + sr0) * p.Cin) * 2 + sc * 16;
    const int w_step = p.Cin * 2 * 32;
    const int cpt = p.Cin / LG_BK;                                     // K chunks per tap block of the weights
    // two register sets: chunk c + 1 (set (c + 1) & 1) is on its way to LDS while chunk c + 2 is still in flight -- one
    // chunk of MFMAs (~0.5 us) does not cover an L2 round trip under load, two do
    u32x4 ra[2][MT], rw[2][4];
    auto issue = [&](auto set, int c) {
        constexpr int S = decltype(set)::value;
        const int tap = c / cpt;
        const int a_soff = __builtin_amdgcn_readfirstlane(c * LG_BK * 2);
        const int w_soff = __builtin_amdgcn_readfirstlane((int)(((long)tap * p.w_tap_stride + (long)(c - tap * cpt) * LG_BK) * 2));
#pragma unroll
        for (int i = 0; i < MT; ++i) ra[S][i] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff + i * a_step, a_soff, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[S][i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff + i * w_step, w_soff, 0);
    };
    const int s_off = sr0 * LG_ROWB + ((sc << 4) ^ lg_swz(sr0));       // rows sr0 + 32 i share the swizzle term of sr0
    // staging stores of one chunk in two halves (behind the two MFMA blocks of an iteration)
    auto store_half = [&](auto set, char* buf, int h) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if ((i & 1) == h) *reinterpret_cast<u32x4*>(buf + s_off + i * 32 * LG_ROWB) = ra[S][i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((i & 1) == h) *reinterpret_cast<u32x4*>(buf + ATILE + s_off + i * 32 * LG_ROWB) = rw[S][i];
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int preA[MT], preW[4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm0 + 16 * i + r16;
        preA[i] = r * LG_ROWB + (lg_swz(r) ^ (kg << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = wn0 + 16 * j + r16;
        preW[j] = ATILE + n * LG_ROWB + (lg_swz(n) ^ (kg << 4));
    }
    auto load = [&](bf16x8 (&a)[MT], bf16x8 (&w)[4], const char* buf, int ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(buf + (preA[i] ^ (ks * 64)));
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(buf + (preW[j] ^ (ks * 64)));
    };
    auto mma = [&](const bf16x8 (&a)[MT], const bf16x8 (&w)[4]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    };

    const int nchunks = p.K / LG_BK;
    const int last = nchunks - 1;
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    issue(S0{}, 0);
    issue(S1{}, last < 1 ? last : 1);
    store_half(S0{}, smem, 0);                                         // chunk 0; chunk 1 stays in set 1
    store_half(S0{}, smem, 1);
    issue(S0{}, last < 2 ? last : 2);
    __syncthreads();
    // Iteration c reads chunk c from LDS buffer c & 1, writes chunk c + 1 (register set (c + 1) & 1, requested two
    // iterations ago) to the other buffer behind the MFMA blocks and requests chunk c + 3 into that same set; loads are
    // unconditional (clamped to the last chunk): a conditional load drains vmcnt at the join.
    auto step = [&](auto land, int c) {
        const char* buf = smem + (c & 1) * BUF;
        char* nxt = smem + ((c + 1) & 1) * BUF;
        bf16x8 aa[MT], wa[4], ab[MT], wb[4];
        load(aa, wa, buf, 0);
        load(ab, wb, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma(aa, wa);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        store_half(land, nxt, 0);                                      // (after the last chunk: a dead buffer)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma(ab, wb);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        store_half(land, nxt, 1);
        issue(land, c + 3 < nchunks ? c + 3 : last);
        __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(S1{}, c);
        if (c + 1 < nchunks) step(S0{}, c + 1);
    }

    lg_epilogue<BM>(p, smem, acc, tid, wm0, wn0, r16, kg, seg, m0, n0, Ms);
}

template <int BM>
static int lingemm_launch(si_ctx* ctx, const LinGemmParams& p, hipStream_t st) {
    const size_t lds = std::max<size_t>(2 * ((size_t)BM * LG_ROWB + LG_WTILE), (size_t)BM * 512);   // operand double buffer / fp32 output image
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(lingemm_kernel<BM>), lds)) return rc;
    const int mtiles = (p.M + BM - 1) / BM;
    // XCD-aware tile order when the whole weight matrix sits comfortably in one XCD's 4 MB L2: every row block's A rows
    // are then fetched into one L2 instead of eight.  Same-box A/B: convolutions -4 ... -9 %, out-proj -5 %, QKV -2 %;
    // FFN1 (4.7 MB of weights, 24 column tiles) +9 % -- there the plain order, which keeps 3 of the 24 weight column
    // tiles resident per XCD, wins -- so the rule is by weight bytes.
    const bool xcd = (double)p.N * p.K * 2.0 <= 3.6e6;
    LinGemmParams q = p;
    const int rows_total = p.nseg * mtiles;
    q.xcd_rows = xcd ? rows_total : 0;
    const unsigned grid = (unsigned)((xcd ? (rows_total + 7) / 8 * 8 : rows_total) * (p.N / LG_BN));
    double rows_real = (double)p.nseg * p.M;                           // ragged batches: the rows that exist
    if (p.seg_m_host) { rows_real = 0; for (int s = 0; s < p.nseg; ++s) rows_real += p.seg_m_host[s]; }
    const double macs = rows_real * p.N * (double)p.K;
    const double outs = rows_real * p.N;
    const double bytes = 2.0 * (rows_real * p.lda + p.nseg * (double)(p.K - p.lda > 0 ? p.K - p.lda : 0)) + outs * ((p.out ? 4 : 0) + (p.out16 ? 2 : 0) + (p.res ? 4 : 0)) + 2.0 * p.N * p.K;
    char name[48];
    snprintf(name, sizeof(name), "lingemm_bf16_%dx128", BM);           // one family per instantiation, as rocprofv3 lists them
    si_prof_begin(ctx, si_prof_shape_name(name, p.M * (long)p.nseg, p.N, p.K), 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(lingemm_kernel<BM>, dim3(grid), dim3(LG_NT), lds, st, q);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller uses the tap-GEMM).
int si_launch_lingemm(si_ctx* ctx, const LinGemmParams& p, hipStream_t st) {
    if (p.N % LG_BN || p.Cin % LG_BK || p.K % LG_BK || p.K != p.ntaps * p.Cin || p.ldo % 4 || p.lda % 8 || p.M <= 0 || p.nseg <= 0) return 1;
    if (p.x_bytes <= 0 || p.w_bytes <= 0 || (long)p.nseg * p.x_seg_stride * 2 + (long)(p.M + 128) * p.lda * 2 >= (1L << 31)) return 1;
    if (!p.out && !p.out16) return si_fail(ctx, SI_EINVAL, "lingemm: no output");
    if (p.res_stats && (!p.res || !p.res_gamma || !p.res_beta || p.nseg != 1)) return si_fail(ctx, SI_EINVAL, "lingemm: a LayerNorm residual needs res, gamma, beta and one flat segment");
    {
        int rc = si_launch_gemmcu(ctx, p, st);                         // one tile per CU where that is one round of the chip (bit-identical results)
        if (rc <= 0) return rc;
        rc = si_launch_gemm256(ctx, p, st);                            // 256 x 256 tiles where they fill the chip (bit-identical results)
        if (rc <= 0) return rc;
    }
    // Tile height: the workgroup slots are 2 per CU; a launch takes ceil(tiles / slots) rounds of a tile's time, which
    // grows with BM (plus a fixed part: prologue, epilogue).  Pick the BM with the smallest rounds x (BM + fixed).
    int bm = 128;
    {
        const long slots = 2L * si_num_cus(ctx);
        double best = 1e30;
        for (int cand : {128, 96, 64}) {
            const long tiles = (p.seg_m_host ? si_vl_tiles(p.seg_m_host, p.nseg, cand) : (long)p.nseg * ((p.M + cand - 1) / cand)) * (p.N / LG_BN);
            const double cost = (double)((tiles + slots - 1) / slots) * (cand + 24);
            if (cost < best * 0.97) { best = cost; bm = cand; }        // ties and near-ties go to the taller tile
        }
    }
    if (bm == 64) return lingemm_launch<64>(ctx, p, st);
    if (bm == 96) return lingemm_launch<96>(ctx, p, st);
    return lingemm_launch<128>(ctx, p, st);
}
