// gemm256.hip -- the encoder's bf16 GEMM on 256 x 256 tiles (gfx950, wave64, v_mfma_f32_16x16x32_bf16, LDS-DMA staging).
//
//     out[seg][m][n] = epi( sum_k A[seg][m][k] * W[n][k] + bias[n] )        (the contract of lingemm.hip: LinGemmParams)
//
// Why a second kernel.  rocprofv3 counters on lingemm's 128 x 128 tiles (profiles/r03_lingemm_pmc.txt) show the vector-memory
// path, not the matrix pipe or LDS, as the limit: 32 KB of operands per 2.1 MFLOP leave L2 for every tile chunk (15.6 KB per
// MFLOP: 5.5 GB per conv1 launch, 45 GB/s per CU against the 66-73 GB/s a CU has been seen to draw from L2), the texture
// addresser is stalled by the L1 45 % of the time, SQ_VMEM_TA_*_FIFO_FULL 18-20 % of the wave cycles, matrix pipe busy 0.29.
// Halving the LDS staging (the operand-from-registers form, lingemm2) or removing erff from the epilogue changed nothing.
// The lever is bytes per flop: a 256 x 256 tile moves 64 KB per 8.4 MFLOP (7.8 KB per MFLOP), each operand byte once per
// workgroup, in full 128-byte lines.
//
// Structure (after the guide's 256^2 "8-phase" GEMM, cdna_hip_programming.md section 5; the schedule below is this file's own):
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 outputs = 8 x 4 MFMA tiles = 128 accumulator registers.
//   * LDS 160 KB: an A ring of 2 K-tiles and a W ring of 3, each K-tile two half-tiles (rows 0-127 / 128-255) of 128 rows x 64 k.
//     Half-tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write); a wave instruction lands
//     1 KB = 8 rows of 128 bytes lane-linearly, so the XOR swizzle (16-byte chunk c of row r at c ^ (((r >> 1) & 3) << 1),
//     conflict-free for the 16x16x32 operand read: tests/test_lds_swizzles.py) is applied on the SOURCE side: lane l fetches
//     the logical chunk that belongs at its physical position.
//   * A K-tile is four phases; a phase = [fragment reads + one half-tile of DMA requests] | barrier | 16 MFMAs (one 64 x 32
//     quadrant of the wave's tile over K = 64) | barrier.  The two wave groups (M half 0 / 1: the two waves of every SIMD)
//     run ONE barrier apart, so while one group's 16 MFMAs occupy the SIMD's matrix pipe its partner issues reads and DMA:
//     the pipe never waits on LDS latency.  Fragments of a K-tile stay in registers (A 2 x 32, W 2 x 16): 24 ds_read_b128
//     per 64 MFMAs.
//         phase 0: read W-sub 0, A-sub 0 ; request W half 0 of tile t + 2 ; MFMA (A0, W0)
//         phase 1: read A-sub 1          ; request W half 1 of tile t + 2 ; MFMA (A1, W0)
//         phase 2: read W-sub 1          ; request A half 0 of tile t + 2 ; MFMA (A1, W1)
//         phase 3:                         request A half 1 of tile t + 2 ; s_waitcnt vmcnt(8) ; MFMA (A0, W1)
//     Every half-tile has a whole K-tile of MFMAs (~2 us) to arrive: with the W halves requested only one tile ahead (a
//     two-deep W ring, 128 KB) the flat M = 6368 launches, whose every tile starts on cold lines, ran 4.2 us per K-tile.
//     Hazards.  WAR: a slot is re-requested at least one whole phase after the phase that last read it, and every read is
//     retired (lgkmcnt(0)) BEFORE the barrier that ends its phase half, in both groups.  RAW: DMA data is ordered for a
//     ds_read only by the requesting waves' vmcnt followed by a barrier the reader has passed: phase 3's vmcnt(8) (all but
//     the four youngest half-tiles: tile t + 1 complete) sits before that phase's first barrier in each group, and tile t + 1
//     is first read after its second.
//   * Epilogue from the accumulators (bias, fast erf-GELU, residual; bf16 through a half trade between lanes l and l + 16).
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

constexpr int G_BM = 256, G_BN = 256, G_BK = 64, G_NT = 512;
constexpr int G_ROWB = G_BK * 2;                                       // 128-byte LDS rows
constexpr int G_HALF = 128 * G_ROWB;                                   // one half-tile: 16 KB
constexpr int G_PAIR = 2 * G_HALF;                                     // both halves of one operand of one K-tile: 32 KB
constexpr int G_WBASE = 2 * G_PAIR;                                    // LDS: A ring of 2 pairs at 0, W ring of 3 pairs behind it: 160 KB

#define G_LDS(off) ((__attribute__((address_space(3))) void*)(smem + (off)))

// Diagnostic build only (-DG256_TIMELINE; tools/exp_encoder_only.py): wall-clock totals of every workgroup's wave 0 in 100 MHz
// ticks, [shape class: 0 = per-clip launches with 24 K-tiles, 1 = N 2304, 2 = N 3072, 3 = other][0 prologue, 1 K loops, 2 epilogues,
// 3 tiles, 4 workgroups, 5 whole kernel].
#ifdef G256_TIMELINE
__device__ unsigned long long g256_tl[4][6];
#define G_TL_DECL const unsigned long long tl_0 = wall_clock64(); unsigned long long tl_t = tl_0, tl_acc[3] = {0, 0, 0}, tl_tiles = 0;
#define G_TL(ph) { const unsigned long long n_ = wall_clock64(); tl_acc[ph] += n_ - tl_t; tl_t = n_; }
#define G_TL_TILE ++tl_tiles;
#define G_TL_FLUSH if (threadIdx.x == 0) { const int c_ = (p.nseg > 1 && p.K == 1536) ? 0 : (p.N == 2304 ? 1 : (p.N == 3072 ? 2 : 3)); for (int q_ = 0; q_ < 3; ++q_) atomicAdd(&g256_tl[c_][q_], tl_acc[q_]); atomicAdd(&g256_tl[c_][3], tl_tiles); atomicAdd(&g256_tl[c_][4], 1ull); atomicAdd(&g256_tl[c_][5], wall_clock64() - tl_0); }
extern "C" int si_debug_g256_timeline(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g256_tl), sizeof(g256_tl)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4][6] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g256_tl), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define G_TL_DECL
#define G_TL(ph)
#define G_TL_TILE
#define G_TL_FLUSH
#endif

__global__ __launch_bounds__(G_NT, 2) void gemm256_kernel(const LinGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // A [2][half 0 | half 1] then W [3][half 0 | half 1], [128 rows][128 bytes] each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // a scalar: the LDS-DMA destinations (M0) are then scalar arithmetic
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kg = lane >> 4;

    const int ntn = p.N / G_BN;
    const int mtiles = (p.M + G_BM - 1) / G_BM;
    // XCD-aware, persistent tile walk.  Workgroup ids b and b + 8 share an L2 (dealt round-robin over the 8 XCDs).  The tiles, in
    // row-block-major order (the column tiles of a row block adjacent), are cut into 8 CONTIGUOUS, equally long runs, one per
    // XCD: a row block's A rows are fetched into one L2 (two at a seam), not eight, and no XCD gets more tiles than another
    // (with whole row blocks per XCD the 25 row blocks of the flat M = 6368 launches put 36 tiles on XCD 0's 32 CUs: two
    // rounds).  Workgroup b takes elements (b >> 3) + i * (gridDim.x >> 3) of its XCD's run; p.xcd_rows = number of tiles;
    // p.persistent = 0: the workgroup stops after its first tile.
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slot_step = gridDim.x >> 3;
    const int run0 = (int)((long)p.xcd_rows * xcd / 8), run1 = (int)((long)p.xcd_rows * (xcd + 1) / 8);
    struct TileRef { int seg, m0, n0, M; bool valid; };
    auto decode = [&](int it) {
        TileRef r{0, 0, 0, p.M, false};
        if (it > 0 && !p.persistent) return r;
        const int tile = run0 + slot0 + it * slot_step;
        if (tile >= run1) return r;
        const int mtx = tile / ntn;
        if (p.seg_m) {                                                 // ragged batches: row blocks numbered segment by segment without gaps
            const SiVlTile v = si_vl_tile(p.seg_m, p.nseg, G_BM, mtx);
            r.seg = v.b; r.m0 = v.row0; r.M = v.L;
        } else {
            r.seg = mtx / mtiles;
            r.m0 = (mtx - r.seg * mtiles) * G_BM;
        }
        r.n0 = (tile - mtx * ntn) * G_BN;
        r.valid = true;
        return r;
    };
    TileRef cur = decode(0);
    if (!cur.valid) return;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x16), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.w_bytes, 0x00020000);
    // ---- LDS-DMA: piece q (0, 1) of a half-tile = rows (2 wave + q) * 8 + (lane >> 3); the lane lands at physical chunk
    // lane & 7 of its row, so it fetches the logical chunk that the swizzle puts there
    const int srow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((srow >> 1) & 3) << 1);
    auto a_base_of = [&](const TileRef& r) { return (int)(((long)r.seg * p.x_seg_stride + (long)(r.m0 + wave * 16 + srow) * p.lda) * 2) + lchunk * 16; };
    auto w_base_of = [&](const TileRef& r) { return (int)(((long)(r.n0 + wave * 16 + srow) * p.Cin) * 2) + lchunk * 16; };
    int a_base = a_base_of(cur), w_base = w_base_of(cur);
    int a_next = 0, w_next = 0;                                        // the same for the workgroup's next tile
    bool has_next = false;
    const int cpt = p.Cin / G_BK;                                      // K-tiles per tap block of the weights
    const int nk = p.K / G_BK;
    const int piece0 = wave * 16 * G_ROWB;                             // LDS offset of this wave's two pieces within a half-tile
    // K-tile kt of the current tile; kt >= nk: K-tile kt - nk of the NEXT tile (its first two K-tiles are requested under the
    // last two of this one, and land while the epilogue runs), or nothing when there is none.  All conditions wave-uniform.
    auto stage_a = [&](int kt, int h, int buf) {
        int base = a_base;
        if (kt >= nk) {
            if (!has_next) return;
            kt -= nk; base = a_next;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int soff = __builtin_amdgcn_readfirstlane(kt * G_BK * 2 + (h * 128 + q * 8) * p.lda * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, G_LDS(buf * G_PAIR + h * G_HALF + piece0 + q * 8 * G_ROWB), 16, base, soff, 0, 0);
        }
    };
    auto stage_w = [&](int kt, int h, int buf) {
        int base = w_base;
        if (kt >= nk) {
            if (!has_next) return;
            kt -= nk; base = w_next;
        }
        const int tap = kt / cpt;
        const long koff = ((long)tap * p.w_tap_stride + (long)(kt - tap * cpt) * G_BK) * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int soff = __builtin_amdgcn_readfirstlane((int)(koff + (long)(h * 128 + q * 8) * p.Cin * 2));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, G_LDS(G_WBASE + buf * G_PAIR + h * G_HALF + piece0 + q * 8 * G_ROWB), 16, base, soff, 0, 0);
        }
    };

    // ---- fragment reads: lane (r16, kg) reads chunk (4 ks + kg) ^ swizzle of row (16 * tile + r16)
    const int foff = r16 * G_ROWB + ((kg << 4) ^ ((((r16 >> 1) & 3) << 1) << 4));
    const int a_slot = wr * G_HALF;                                    // this wave's A half
    const int w_slot = G_WBASE + (wc >> 1) * G_HALF + (wc & 1) * 64 * G_ROWB;   // its 64 W rows inside their half
    bf16x8 fa[2][4][2], fw[2][2][2];
    auto read_a = [&](int mq, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fa[mq][i][ks] = *reinterpret_cast<const bf16x8*>(smem + buf * G_PAIR + a_slot + (mq * 64 + i * 16) * G_ROWB + (foff ^ (ks * 64)));
    };
    auto read_w = [&](int nq, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fw[nq][j][ks] = *reinterpret_cast<const bf16x8*>(smem + buf * G_PAIR + w_slot + (nq * 32 + j * 16) * G_ROWB + (foff ^ (ks * 64)));
    };
    f32x4 acc[8][4];
    auto mma = [&](int mq, int nq) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[mq * 4 + i][nq * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nq][j][ks], fa[mq][i][ks], acc[mq * 4 + i][nq * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase's first half: every LDS read retired, then the barrier; `drain` >= 0: first the DMA of the next K-tile
    auto close_loads = [&](int drain) {
        __builtin_amdgcn_sched_barrier(0);
        if (drain == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (drain == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto close_mma = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // K-tile t: A from ring slot t % 2, W from ring slot t % 3.  Its four phases request ALL of K-tile t + 2: the W halves into
    // the W slot that tile t - 1 read last, the A halves (phases 2, 3) into this tile's own A slot, whose last reads are in phase 1.
    auto ktile = [&](auto abuf, auto wbuf, int t) {
        constexpr int AB = decltype(abuf)::value, WB = decltype(wbuf)::value, WN = (WB + 2) % 3;
        // requests past the end of the work are skipped: the counted wait then has to drain
        const int drain = (t + 2 < nk || has_next) ? 8 : 0;
        read_w(0, WB); read_a(0, AB); stage_w(t + 2, 0, WN); close_loads(-1); mma(0, 0); close_mma();
        read_a(1, AB);                stage_w(t + 2, 1, WN); close_loads(-1); mma(1, 0); close_mma();
        read_w(1, WB);                stage_a(t + 2, 0, AB); close_loads(-1); mma(1, 1); close_mma();
                                      stage_a(t + 2, 1, AB); close_loads(drain); mma(0, 1); close_mma();
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;

    // ---- prologue: K-tiles 0 and 1 of the first tile requested; tile 0 landed and visible
    G_TL_DECL
    stage_a(0, 0, 0); stage_a(0, 1, 0); stage_w(0, 0, 0); stage_w(0, 1, 0);
    stage_a(1, 0, 1); stage_a(1, 1, 1); stage_w(1, 0, 1); stage_w(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                         // the second wave group runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    const bool gelu = p.act == SI_ACT_GELU;
    const bool has_res = p.res != nullptr;
    const bool res_ln = p.res_stats != nullptr;
    G_TL(0)
    for (int it = 0;; ++it) {
        const TileRef nxt = decode(it + 1);
        has_next = nxt.valid;
        if (has_next) { a_next = a_base_of(nxt); w_next = w_base_of(nxt); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (a persistent walk needs nk % 6 == 0, so that every tile starts on ring slots 0 / 0: the launcher's condition)
        for (int t = 0; t < nk; t += 6) {
            ktile(I0{}, I0{}, t);
            if (t + 1 < nk) ktile(I1{}, I1{}, t + 1);
            if (t + 2 < nk) ktile(I0{}, I2{}, t + 2);
            if (t + 3 < nk) ktile(I1{}, I0{}, t + 3);
            if (t + 4 < nk) ktile(I0{}, I1{}, t + 4);
            if (t + 5 < nk) ktile(I1{}, I2{}, t + 5);
        }

        G_TL(1)
        // ---- epilogue from the accumulators: lane (r16, kg) holds row 16 mi + r16, columns 16 nj + 4 kg + [0, 4) of the wave's tile.
        // (Conditions hoisted out of the unrolled loops: a per-element "load or zero" makes hipcc branch around every load and
        // wait vmcnt(0) behind each -- 32 dependent L2 round trips per lane.)  The next tile's first K-tiles are in flight.
        const long obase = (long)cur.seg * p.o_seg_stride;
        const int ncol0 = cur.n0 + wc * 64;
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + 16 * j + 4 * kg);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = cur.m0 + wr * 128 + 16 * i + r16;
            const bool live = m < cur.M;
            const long orow = obase + (long)(live ? m : cur.M - 1) * p.ldo + ncol0;  // dead rows read row M - 1 and store nothing
            f32x4 rv[4];
            if (has_res) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const f32x4*>(p.res + orow + 16 * j + 4 * kg);
                if (res_ln) {                                          // the residual is LayerNorm(res): si_ln_apply on the row's (mean, rstd)
                    const f32x2 ms = *reinterpret_cast<const f32x2*>(p.res_stats + 2 * (long)(live ? m : cur.M - 1));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 lg = *reinterpret_cast<const f32x4*>(p.res_gamma + ncol0 + 16 * j + 4 * kg);
                        const f32x4 lb = *reinterpret_cast<const f32x4*>(p.res_beta + ncol0 + 16 * j + 4 * kg);
#pragma unroll
                        for (int e = 0; e < 4; ++e) rv[j][e] = si_ln_apply(rv[j][e], ms[0], ms[1], lg[e], lb[e]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
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
                // bf16: lanes l and l + 16 trade halves of a column-tile pair (v_permlane16_swap): an even kg then owns columns
                // 4 kg + [0, 8) of tile 2 t, an odd one columns 4 (kg - 1) + [0, 8) of tile 2 t + 1: 16 bytes per lane, 64 per row
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[i][2 * t], bf16x4));
                    u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[i][2 * t + 1], bf16x4));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const auto r = __builtin_amdgcn_permlane16_swap(p0[q], p1[q], false, false);
                        p0[q] = r[0]; p1[q] = r[1];
                    }
                    const int col = 16 * (2 * t + (kg & 1)) + 4 * (kg & ~1);
                    if (live) *reinterpret_cast<u32x4*>(p.out16 + orow + col) = u32x4{p0[0], p0[1], p1[0], p1[1]};
                }
            }
        }
        G_TL(2)
        G_TL_TILE
        if (!has_next) break;
        cur = nxt; a_base = a_next; w_base = w_next;
    }
    G_TL_FLUSH
    if (wr == 0) __builtin_amdgcn_s_barrier();                         // rejoin the groups (equal barrier counts)
}

// SI_OK when launched, negative on error, 1 when the shape is not covered or the rule leaves it to the 128-row kernels.
int si_launch_gemm256(si_ctx* ctx, const LinGemmParams& p, hipStream_t st) {
    if (p.N % G_BN || p.Cin % G_BK || p.K % G_BK || p.K != p.ntaps * p.Cin || p.K < 2 * G_BK || p.ldo % 4 || p.lda % 8 || p.M <= 0 || p.nseg <= 0) return 1;
    if (p.x_bytes <= 0 || p.w_bytes <= 0 || (long)p.nseg * p.x_seg_stride * 2 + (long)(p.M + 256) * p.lda * 2 >= (1L << 31)) return 1;
    if ((long)(p.N + 256) * p.Cin * 2 + (long)p.ntaps * p.w_tap_stride * 2 >= (1L << 31)) return 1;
    const int mtiles = (p.M + G_BM - 1) / G_BM;
    const long row_blocks = p.seg_m_host ? si_vl_tiles(p.seg_m_host, p.nseg, G_BM) : (long)p.nseg * mtiles;   // ragged batches: no gaps
    double rows_real = (double)p.nseg * p.M;
    if (p.seg_m_host) { rows_real = 0; for (int s = 0; s < p.nseg; ++s) rows_real += p.seg_m_host[s]; }
    if (p.seg_m && !p.seg_m_host) return 1;
    if (p.res_stats && p.nseg != 1) return 1;                          // a LayerNorm residual: one flat segment
    // The results are bit-identical to lingemm's (same K order, same MFMA and operand roles, same epilogue), so the choice is
    // purely one of speed and may depend on the batch: 256-row tiles with one workgroup per CU pay when the tiles fill whole
    // rounds of the chip's CUs and few of their rows are padding.  (B = 32 x 4 s, HuBERT-base: the first four strided
    // convolutions and the QKV projection; FFN1's 300 tiles are 1.17 rounds and stay on the 128-row kernel.)
    const int opt = si_opt_gemm256(ctx);
    if (opt == 0) return 1;
    if (si_num_cus(ctx) < 8) return 1;                                 // the XCD run split needs a grid that is a positive multiple of 8 (partitioned devices: lingemm)
    if (opt == 1) {
        const int cus = si_num_cus(ctx);
        const long tiles = row_blocks * (p.N / G_BN);
        const double fill = (double)tiles / (double)((tiles + cus - 1) / cus * cus);
        const double rows = rows_real / ((double)row_blocks * G_BM);
        if (fill * rows < 0.72) return 1;
    }
    const size_t lds = 5 * (size_t)G_PAIR;                             // all 160 KB of a CU's LDS
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(gemm256_kernel), lds)) return rc;
    LinGemmParams q = p;
    const int tiles = (int)(row_blocks * (p.N / G_BN));
    q.xcd_rows = tiles;
    // One workgroup per CU; with more tiles than CUs a workgroup walks its XCD's run and requests the next tile's first K-tiles
    // under the last two of the current one.  Tiles must then start on ring slots 0 / 0: K a multiple of 6 K-tiles; otherwise
    // one workgroup per tile.
    const int run_max = (tiles + 7) / 8;                               // longest run of one XCD
    const int nk = p.K / G_BK;
    const int cus = si_num_cus(ctx);
    q.persistent = (nk % 6 == 0 && run_max * 8 > cus) ? 1 : 0;
    const unsigned grid = (unsigned)(q.persistent ? std::min(run_max, cus / 8) * 8 : run_max * 8);
    const double macs = rows_real * p.N * (double)p.K;
    const double outs = rows_real * p.N;
    const double bytes = 2.0 * (rows_real * p.lda + p.nseg * (double)(p.K - p.lda > 0 ? p.K - p.lda : 0)) + outs * ((p.out ? 4 : 0) + (p.out16 ? 2 : 0) + (p.res ? 4 : 0)) + 2.0 * p.N * p.K;
    si_prof_begin(ctx, si_prof_shape_name("gemm256_bf16", p.M * (long)p.nseg, p.N, p.K), 2.0 * macs, bytes, st);
    hipLaunchKernelGGL(gemm256_kernel, dim3(grid), dim3(G_NT), lds, st, q);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
