// respair.hip -- one ResBlock1 step  y' = y + conv2(lrelu(conv1(lrelu(y))))  as ONE kernel for the NARROW vocoder stages
// (C = 32 / 64 channels) in the fp16 activation-stream mode (gfx950, wave64, v_mfma_f32_16x16x32_f16).
//
// I_ea/hifi_gan/models.py:36-43 runs, per (resblock, dilation): xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt);
// x = xt + x.  With N = C <= 64 one workgroup owns ALL channels of its rows, so the pair's intermediate stays in LDS
// (respair_wide.hip is the same kernel for C = 128 / 256; the description of the phases is there).  What differs here:
//   * a K step of the MFMA (32 channels) is half or all of C, so the unit of weight streaming is a GROUP OF TAPS, not a
//     channel chunk: C = 32 holds all taps of a convolution in one slab (both convolutions of the pair are resident
//     after the prologue: no streaming at all), C = 64 streams slabs of 4 taps (32 KB) through the double buffer.  The
//     round-1 kernel re-staged a C x C slab and crossed a workgroup barrier for EVERY tap -- 4 to 16 MFMAs per barrier;
//   * C = 32: 256 rows per 4-wave workgroup and 65 KB of LDS, two workgroups per CU (one's tile load and output pass
//     overlap the other's MFMAs); C = 64: 512 rows per 8-wave workgroup;
//   * LDS rows are 64 / 128 bytes: the chunk swizzle term of a row keeps chunk bit 0 (see swz16 in respair_wide.hip) and
//     spreads the 2 or 4 rows of a 256-byte bank period.
// Arithmetic is that of the two-launch tap-GEMM form in the same mode: fp16 operands, fp32 accumulate, the intermediate
// rounded to fp16 once, (acc + y) * alpha (+ previous) in fp32 with the accumulators started from the bias, one
// saturating rounding on store.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int RPN_HALO = 50;                                          // (k - 1) * dil <= 50: k = 11, dil = 5

// (chunk XOR term of a row) << 4 for 64- and 128-byte rows: chunk c of row r lives at chunk c ^ ((r >> 1) & 3) (64-byte rows:
// 4 chunks) / c ^ (r & 7) (128-byte rows: 8 chunks).  Checked by enumeration over the hardware's lane groups: the
// 16x16x32 operand read (ds_read_b128) is conflict-free for ANY first row, so are the staging stores, and so are the
// epilogue's 16-byte stores of 8 consecutive rows to one logical chunk (the first swizzle of this round kept chunk bit 0
// fixed and left those 2-way conflicted).
template <int ROWB>
__device__ __forceinline__ int rpn_swz(int row) {
    if constexpr (ROWB == 128) return (row & 7) << 4;
    else return ((row >> 1) & 3) << 4;
}

// ACC: the launch adds into the previous contents of out16.  Without it the registers of the accumulate rows are free and
// the NEXT tile's activation rows are requested before the last slab (respair_wide.hip).
// VL: ragged batches (respair_wide.hip)
template <int C, int R1, int WARPS_M, int TPS, bool ACC, bool VL>
__global__ __launch_bounds__(64 * WARPS_M, 2) void respair_kernel(const ResPairParams p) {
    static_assert(R1 == WARPS_M * 64 && (C == 32 || C == 64), "64-row wave tiles over all C channels");
    constexpr int NT = 64 * WARPS_M;
    constexpr int ROWB = C * 2;                                        // bytes per activation / intermediate / weight row
    constexpr int ROWBO = C * 4;                                       // bytes per fp32 output-image row
    constexpr int CPR = C / 8;                                         // 16-byte chunks per row
    constexpr int KS = C / 32;                                         // MFMA k-steps (K = 32) per tap
    constexpr int TN = C / 16;                                         // 16-column tiles per wave
    // C = 64: the activation / intermediate rows are PADDED to 160 bytes and not swizzled (enumerated over the hardware's lane
    // groups: operand reads and staging stores stay conflict-free, the traded epilogue stores become 2-way), so a tap's
    // row offset is a plain scalar add and the four row tiles of a wave sit at immediate offsets: 1 VALU op per k-step
    // where the XOR swizzle took 4 per fragment.  C = 32 keeps the swizzled 64-byte rows (two workgroups per CU must fit).
    constexpr int YPAD = C == 64 ? 32 : 0;
    constexpr int YS = ROWB + YPAD;                                    // bytes from one activation row to the next
    constexpr int YBYTES = (R1 + RPN_HALO) * YS;
    constexpr int WBYTES = TPS * C * ROWB;                             // one slab: TPS taps x [C n][C ci]
    constexpr int WSLOTS = (WBYTES / 16 + NT - 1) / NT;
    constexpr int YRPP = NT / CPR;                                     // rows per staging pass
    constexpr int YSLOTS = (R1 + RPN_HALO + YRPP - 1) / YRPP;
    constexpr int OXM = (C / 4 - 1) < 15 ? (C / 4 - 1) : 15;           // chunk XOR mask of the output image
    static_assert(R1 * ROWBO <= YBYTES + 2 * WBYTES, "the output image reuses the operand region (not the biases behind it)");

    // MODE.FP16_OVFL (hwreg 1, bit 23): a conversion to fp16 that overflows gives +-65504 instead of +-inf, i.e. the
    // clamp before every rounding comes with the conversion
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ys = smem;                                             // [R1 + 50][C] fp16: lrelu(y), later t (rows < R1)
    char* const Ws = smem + YBYTES;                                    // [2][TPS][C][C] fp16 weight slabs
    float* const Bs = reinterpret_cast<float*>(smem + YBYTES + 2 * WBYTES);   // [2][C] fp32: b1, b2

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int wm0 = wave * 64;
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

    // ---- weight slabs: slab s = taps [TPS * q, TPS * (q + 1)) of conv 1 (s < NS1) or conv 2 (q = s - NS1); the weights are
    //      [tap][n][ci], so a slab is one contiguous run; taps past k read as zero through the descriptor
    const int NS1 = (k + TPS - 1) / TPS, NS = 2 * NS1;
    u32x4 rw[WSLOTS];
    auto issueW = [&](int s) {
        const bool second = s >= NS1;
        const int soff = __builtin_amdgcn_readfirstlane((second ? s - NS1 : s) * WBYTES);
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i)
            rw[i] = __builtin_amdgcn_raw_buffer_load_b128(second ? w2rsrc : w1rsrc, (tid + i * NT) * 16, soff, 0);
    };
    auto storeW = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) {
            const int q = tid + i * NT;                                // 16-byte chunk of the slab: row q / CPR, chunk q % CPR
            if (WBYTES / 16 % NT == 0 || q < WBYTES / 16) {
                const int r = q / CPR, c = q - r * CPR;
                *reinterpret_cast<u32x4*>(dst + r * ROWB + ((c << 4) ^ rpn_swz<ROWB>(r))) = rw[i];
            }
        }
    };

    // the halo'd activation rows of a tile -> registers (rows outside the clip read as zero through the clip's descriptor)
    const int yc = tid % CPR, yr0 = tid / CPR;
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

    issueW(0);
    issueY(blockIdx.x);
    if (tid < C / 2) {                                                 // biases -> LDS: the epilogues read them per lane
        const int which = tid / (C / 4), c4 = (tid % (C / 4)) * 4;
        *reinterpret_cast<f32x4*>(Bs + which * C + c4) = *reinterpret_cast<const f32x4*>((which ? p.b2 : p.b1) + c4);
    }

  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int nxt = tile + (int)gridDim.x < total ? tile + (int)gridDim.x : tile;   // clamped: the loads below stay unconditional
    int b, m0, Lb;                                                     // clip, first output row of this tile, the clip's rows
    tile_of(tile, b, m0, Lb);
    const int t_row0 = m0 - p2;                                        // clip row of intermediate row 0
    const long seg = (long)b * p.L * C;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.y16 + seg), 0, Lb * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out16 + seg, 0, Lb * C * 2, 0x00020000);

    // ---- slab 0 and the activation tile (both requested one tile ago, or at kernel entry) -> LDS: raw fp16 -> leaky-ReLU(0.1)
    //      on the packed halves (rows outside the clip read as zero)
    storeW(Ws);
    issueW(1);                                                         // slab 1 (NS >= 2) goes to the other buffer behind slab 0's MFMAs
#pragma unroll
    for (int i = 0; i < YSLOTS; ++i) {
        const int r = yr0 + i * YRPP;
        if ((i + 1) * YRPP <= R1 || r < R0) {                           // rows < R1 always exist: no branch around their loads
            f16x8 h = __builtin_bit_cast(f16x8, ry[i]);
            h = __builtin_elementwise_max(h, h * (_Float16)0.1f);     // leaky-ReLU(0.1) = max(x, 0.1 x), packed
            *reinterpret_cast<f16x8*>(Ys + r * YS + (YPAD ? yc << 4 : (yc << 4) ^ rpn_swz<ROWB>(r))) = h;
        }
    }
    __syncthreads();

    f32x4 acc[4][TN];                                                  // [time tile i][channel tile j], transposed 16 x 16 tiles
    // the accumulators start from the bias of this lane's four consecutive channels: no add per element in the epilogues
    auto init_acc = [&](const float* bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * j + 4 * kg);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = bv;
        }
    };
    init_acc(Bs);
    // row tile i of the activations is 16 * i rows (an immediate) behind row tile 0; channel tile j of a weight slab is 16 * j
    // rows behind tile 0 with the same swizzle term (16 rows keep it), and k-step 1 flips byte bit 6 of the swizzled chunk
    const int lin0 = (wm0 + r16) * YS + (kg << 4);
    int preW[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) preW[ks] = r16 * ROWB + (rpn_swz<ROWB>(r16) ^ (kg << 4) ^ (ks * 64));
    // one slab = `ntap` taps x KS k-steps; step q = tap * KS + ks.  Fragments are double-buffered in registers: the reads
    // of step q + 1 are issued before the MFMAs of step q (clamped to the last step: a harmless re-read).
    auto compute = [&](int tap0, bool second, int ntap, const char* Wc) {
        const int nsteps = ntap * KS;
        auto load = [&](f16x8 (&y)[4], f16x8 (&w)[TN], int q, const int ks) {   // ks == q % KS, passed as a literal
            const int tl = KS == 1 ? q : q >> 1;
            // linear byte offset of (row, k group) plus the step's scalar part; on swizzled rows (C = 32) the swizzle is an XOR
            // of address bits (row bits 1-2 -> byte bits 4-5 on 64-byte rows).  The VALU shares the SIMD's issue port with the
            // MFMAs, so every address op here is paid for in MFMA issue slots.
            const int soff = (second ? tap0 + tl : (tap0 + tl) * d) * YS + ks * 64;
            const int lin = lin0 + soff;
            // (swizzled rows: 16 rows further is 1024 bytes further, above every bit the swizzle touches -- one swizzled address
            //  and immediate offsets, as in reschain.hip)
            const char* yp = Ys + (YPAD ? lin : lin ^ ((lin >> 3) & 0x30));
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const f16x8*>(yp + i * (16 * YS));
            const char* wp = Wc + tl * (C * ROWB) + preW[ks];
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const f16x8*>(wp + j * (16 * ROWB));
        };
        auto mma = [&](const f16x8 (&y)[4], const f16x8 (&w)[TN]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], y[i], acc[i][j], 0, 0, 0);
        };
        f16x8 ya[4], wa[TN], yb[4], wb[TN];
        load(ya, wa, 0, 0);
        for (int q = 0; q < nsteps; q += 2) {
            load(yb, wb, q + 1 < nsteps ? q + 1 : nsteps - 1, KS - 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma(ya, wa);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            load(ya, wa, q + 2 < nsteps ? q + 2 : nsteps - KS, 0);         // (an even step: the k-step stays a compile-time constant)
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < nsteps) {
                __builtin_amdgcn_s_setprio(1);
                mma(yb, wb);
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Slab s is read from buffer s & 1.  Slab s + 1 -- requested during slab s - 1 -- is written to the other buffer behind
    // slab s's MFMAs (that buffer was last read during slab s - 1; every wave has passed the barrier that ended it), then
    // slab s + 2 is requested.
    for (int s = 0; s < NS; ++s) {
        const bool second = s >= NS1;
        const int q = second ? s - NS1 : s;
        const int tap0 = q * TPS;
        if (s == NS - 1) {
            // last slab: the weight registers are free (slab 0 of the next tile) and, without an accumulate operand, so are
            // the activation registers: the next tile travels under this slab's MFMAs.  (nxt == tile at the end: harmless.)
            issueW(0);
            if constexpr (!ACC) issueY(nxt);
        }
        compute(tap0, second, min(TPS, k - tap0), Ws + (s & 1) * WBYTES);
        if (s == NS1 - 1) {
            __syncthreads();                                           // every wave has finished reading the activation tile
            // ---- phase-1 epilogue: leaky-ReLU (the bias is in the accumulators), zero outside the clip, fp16, over the activation
            //      tile, as 16-byte stores after lanes l and l + 16 traded chunk halves (respair_wide.hip, reschain.hip: the
            //      8-byte stores of the accumulator layout are 4-way bank conflicted on these rows)
            // (a tile whose intermediate rows all lie inside the clip -- all but the first and last of a clip -- skips the factor:
            //  the branch is uniform and the epilogue's VALU work is time no MFMA overlaps)
            auto epi1 = [&](auto edge_tag) {
                constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                float inside[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int grow = t_row0 + wm0 + 16 * (2 * ip + u) + r16;
                    inside[u] = (!EDGE || (grow >= 0 && grow < Lb)) ? 1.f : 0.f;   // as a factor: no branch per element
                }
                const int ms = wm0 + 16 * (2 * ip + (kg & 1)) + r16;           // the row this lane stores after the trade
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    u32x2 pk[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f16x4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(EDGE ? si_lrelu01(acc[2 * ip + u][j][e]) * inside[u] : si_lrelu01(acc[2 * ip + u][j][e]));   // saturating (MODE.FP16_OVFL)
                        pk[u] = __builtin_bit_cast(u32x2, hv);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const auto r = __builtin_amdgcn_permlane16_swap(pk[0][q], pk[1][q], false, false);
                        pk[0][q] = r[0]; pk[1][q] = r[1];
                    }
                    const int ch = 2 * j + (kg >> 1);                          // 16-byte chunk (8 channels) of the row
                    *reinterpret_cast<u32x4*>(Ys + ms * YS + (YPAD ? ch << 4 : (ch << 4) ^ rpn_swz<ROWB>(ms))) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                }
            }
            };
            if (t_row0 >= 0 && t_row0 + R1 <= Lb) epi1(std::false_type{});
            else epi1(std::true_type{});
            init_acc(Bs + C);
        }
        if (s + 1 < NS) {
            storeW(Ws + ((s + 1) & 1) * WBYTES);
            if (s + 2 < NS) issueW(s + 2);
        }
        __syncthreads();                                               // slab s + 1 (and, after phase 1, the intermediate) is visible
    }

    // ---- final epilogue: accumulators -> fp32 image of the output tile in LDS -> row-contiguous residual add + store.
    //      A lane owns 8 consecutive channels of OPASS output rows; its residual / accumulate rows are requested first.
    constexpr int ORPP = NT / CPR;
    constexpr int OPASS = R1 / ORPP;
    const int c8 = tid % CPR, or0 = tid / CPR;
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = wm0 + 16 * i + r16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = 4 * j + kg;                                 // 16-byte chunk (4 channels) of the output row
            *reinterpret_cast<f32x4*>(smem + m * ROWBO + ((co ^ (m & OXM)) << 4)) = acc[i][j];
        }
    }
    __syncthreads();
    {
#pragma unroll
        for (int it = 0; it < OPASS; ++it) {
            const int o = or0 + it * ORPP;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + o * ROWBO + (((2 * c8) ^ (o & OXM)) << 4));
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + o * ROWBO + (((2 * c8 + 1) ^ (o & OXM)) << 4));
            const f16x8 rh = __builtin_bit_cast(f16x8, res[it]);
            f16x8 ph = {};
            if constexpr (ACC) ph = __builtin_bit_cast(f16x8, prev[it]);
            f16x8 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = e < 4 ? a0[e] : a1[e - 4];
                float v = (a + (float)rh[e]) * p.alpha;
                if constexpr (ACC) v += (float)ph[e];
                out[e] = (_Float16)v;                                  // saturating (MODE.FP16_OVFL)
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, out), orsrc, goff[it], 0, 0);
        }
    }
    if constexpr (ACC) issueY(nxt);                                    // (no registers to spare earlier in this variant)
    __syncthreads();                                                   // the output image is consumed: the next tile may stage into LDS
  }
}

template <int C, int R1, int WARPS_M, int TPS>
static int respair_launch(si_ctx* ctx, const ResPairParams& p, hipStream_t st) {
    const int BMo = R1 - (p.k - 1);
    const size_t lds = (size_t)(R1 + RPN_HALO) * (C * 2 + (C == 64 ? 32 : 0)) + 2 * (size_t)TPS * C * C * 2 + 2 * (size_t)C * 4;
    const bool vl = p.lens != nullptr;
    auto kern = vl ? (p.accumulate ? respair_kernel<C, R1, WARPS_M, TPS, true, true> : respair_kernel<C, R1, WARPS_M, TPS, false, true>)
                   : (p.accumulate ? respair_kernel<C, R1, WARPS_M, TPS, true, false> : respair_kernel<C, R1, WARPS_M, TPS, false, false>);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    // persistent workgroups: as many as are resident at once (two per CU for the 4-wave C = 32 form), each walking tiles
    const int total = vl ? (int)si_vl_tiles(p.lens_host, p.B, BMo) : ((p.L + BMo - 1) / BMo) * p.B;
    if (total <= 0) return SI_OK;
    ResPairParams pk = p;
    pk.total_tiles = total;
    const int grid = std::min(total, si_num_cus(ctx) * (WARPS_M == 4 ? 2 : 1));
    char name[48];
    snprintf(name, sizeof(name), p.accumulate ? "respair_f16_c%d_acc" : "respair_f16_c%d", C);   // one family per instantiation
    double rows = (double)p.B * p.L;
    if (vl) { rows = 0; for (int b = 0; b < p.B; ++b) rows += p.lens_host[b]; }
    const double elems = rows * C;
    si_prof_begin(ctx, name, 2.0 * 2.0 * elems * C * p.k, elems * (2.0 + 2.0 + (p.accumulate ? 2.0 : 0.0)) + 2.0 * 2.0 * p.k * C * C, st);   // y read once (it is also the residual), out written [, previous out read]
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WARPS_M), lds, st, pk);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller launches the two convolutions).
int si_launch_respair(si_ctx* ctx, int C, const unsigned short* y16, unsigned short* out16, const void* w1, const void* w2,
                      const float* b1, const float* b2, int B, int L, int k, int dil, float alpha, int accumulate, hipStream_t st,
                      const int32_t* lens, const int32_t* lens_host, float out_slope) {
    if ((C != 32 && C != 64 && C != 128 && C != 256) || k < 3 || k > 11 || (k & 1) == 0 || (k - 1) * dil > 50 ||
        ((long)L + 1024) * C * 2 >= (1L << 31)) return 1;
    if (!b1 || !b2) return 1;
    if ((lens == nullptr) != (lens_host == nullptr)) return si_fail(ctx, SI_EINVAL, "respair: ragged batches need the lengths on the device and on the host");
    ResPairParams p{y16, out16, static_cast<const unsigned short*>(w1), static_cast<const unsigned short*>(w2), b1, b2, B, L, k, dil, alpha, accumulate, lens, lens_host, 0, out_slope};
    if (out_slope != 1.f && C < 128) return 1;                         // (only the wide kernel applies it)
    if (C >= 128) return si_launch_respair_wide(ctx, C, p, st);
    // C = 32: every tap of a convolution in one slab (11 x 2 KB), 256 rows, two 4-wave workgroups per CU
    // C = 64: 4 taps per slab (32 KB), 512 rows, one 8-wave workgroup per CU
    // (C = 64 as two 4-wave workgroups per CU on 256 rows with 2-tap slabs, same-box A/B: 153 / 235 / 317 us per launch for
    // k = 3 / 7 / 11 against 160 / 231 / 293 -- kept on 512 rows; again after the tap loop lost its address arithmetic:
    // 190 us per launch over the family against 178)
    return C == 32 ? respair_launch<32, 256, 4, 11>(ctx, p, st) : respair_launch<64, 512, 8, 4>(ctx, p, st);
}
