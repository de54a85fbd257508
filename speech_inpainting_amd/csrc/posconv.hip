// posconv.hip -- HuBERT's positional convolution in the bf16 encoder mode (gfx950, wave64, v_mfma_f32_16x16x32_bf16).
//
// modeling_hubert.py:45-103,439-440: h2 = h + GELU(conv(h) + b), conv = weight-normed Conv1d(H -> H, k = 128, padding 64, groups 16)
// with the last output frame dropped (HubertSamePadLayer).  Per group g: out[t][co] = sum_{tap < 128} sum_{ci < Cg} x[t + tap - 64][g Cg + ci]
// * w[g][tap][co][ci], Cg = 48 (base) / 64 (large): a GEMM with N = Cg columns and K = 128 Cg whose A rows OVERLAP (row t + 1 is row t
// shifted by Cg elements).  The generic tap-GEMM ran it on 256 x 64 tiles -- one clip's 199 rows in 256, 48 columns in 64: 58 % useful
// MFMA work -- and streamed the group's 590 KB of weights once per tile: 512 tiles x 590 KB = 302 MB per launch, 146 us at 0.16 of the
// bf16 peak (round 3: rocprofv3 counter traffic 375 MB against 68 MB algorithmic).  Here:
//   * one 8-wave workgroup = one group x TWO clips (clips of at most 256 frames) or x a 512-row block of one clip: 32 row tiles of
//     16; N = Cg exactly (3 / 4 MFMA column tiles, no padding columns); the group's weights stream through LDS once per workgroup
//     (256 workgroups x 590 KB = 151 MB for 32 x 4 s clips);
//   * the clips' rows (+ 127 halo rows, zero outside the clip: the conv's padding) are staged ONCE into LDS as bf16 -- every tap is
//     then a row offset into that tile, and a k-step of 32 flattened (tap, ci) elements is, per lane, one 16-byte read inside one
//     row (Cg % 8 == 0); 96- / 160-byte rows: conflict-free operand reads for every first row and tap phase (enumerated);
//   * weights: chunks of 4 k-steps (128 flattened K x Cg rows) through a double buffer, global -> registers a chunk ahead -> LDS,
//     one barrier per chunk = 48 / 64 MFMAs per wave; read straight from the tap-GEMM's packed layout W[g][tap][n][ci];
//   * orientation D^T = W x X^T (lingemm.hip): a lane holds one frame and four consecutive channels: 16-byte residual reads / stores.
// Arithmetic: bf16 operands (h rounded while staging, as the tap-GEMM did), fp32 accumulation in (tap, ci) order, exact erf-GELU,
// fp32 residual: the tap-GEMM's results up to the order of the fp32 sum.  Ragged batches: per-clip row offset + frame count (device
// arrays); a clip's result does not depend on which other clip shares its workgroup.
#include <algorithm>
#include <cstdio>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int PC_NT = 512;                 // 8 waves
constexpr int PC_KC = 4;                   // k-steps (of 32) per weight chunk
constexpr int PC_HALF_ROWS = 256;          // output rows per half of the workgroup's tile (16 row tiles of 16)
}  // namespace

struct PosConvParams {
    const float* x;                // (rows, H) fp32: h
    float* out;                    // (rows, H) fp32: h + gelu(conv(h) + b)
    const unsigned short* w;       // packed W[g][tap][Npad][Cg] bf16
    const float* bias;             // [H]
    const int32_t* row_off;        // (B) first row of each clip, or null: b * T
    const int32_t* lens;           // (B) frames of each clip, or null: T
    int B, T, H, groups, ntaps, Npad, pad;
    int pair;                      // 1: a workgroup takes clips 2 j and 2 j + 1 (all clips <= 256 frames); 0: 512-row blocks of one clip
    int blocks_per_clip;           // pair == 0: ceil(Tmax / 512)
};

__device__ __forceinline__ float pc_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int CG>
__global__ __launch_bounds__(PC_NT, 1) void posconv_kernel(const PosConvParams p) {
    constexpr int NTC = CG / 16;                                       // MFMA column tiles
    // bytes per LDS row of the activation tile / per weight row of a chunk: by enumeration over the hardware's ds_read_b128 lane groups
    // (tools/lds_conflicts.py) 96 (unpadded) and 160 are conflict-free for the operand read at any first row and tap phase, 288 for the
    // weight rows; the first choice here (+16 bytes: 112 / 144 / 272) was 2-way conflicted everywhere -- PMC: 47 % of the kernel's LDS
    // cycles were bank conflicts
    constexpr int RS = CG == 48 ? 96 : 160;
    constexpr int XROWS = PC_HALF_ROWS + 128;                          // rows per half: 256 outputs + 127 halo (+1)
    constexpr int XBYTES = XROWS * RS;
    constexpr int WROW = PC_KC * 64 + 32;                              // bytes per weight row of a chunk (4 k-steps x 64 B + pad)
    constexpr int WBYTES = CG * WROW;
    constexpr int WSLOTS = (CG * PC_KC * 4 + PC_NT - 1) / PC_NT;       // 16-byte pieces of a chunk per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Xs = smem;                                             // [2][XROWS][RS]
    char* const Ws = smem + 2 * XBYTES;                                // [2][CG][WROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int g = blockIdx.y;
    const int hsel = wave >> 2;                                        // which half of the tile this wave works on
    // ---- the two halves: (clip, first row) each
    int clip[2], r0[2], Tc[2];
    long grow0[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int c, rr;
        if (p.pair) { c = 2 * blockIdx.x + h; rr = 0; }
        else { c = blockIdx.x / p.blocks_per_clip; rr = (blockIdx.x % p.blocks_per_clip) * 2 * PC_HALF_ROWS + h * PC_HALF_ROWS; }
        const bool ok = c < p.B;
        clip[h] = ok ? c : 0;
        Tc[h] = ok ? (p.lens ? p.lens[clip[h]] : p.T) : 0;
        r0[h] = rr;
        grow0[h] = p.row_off ? (long)p.row_off[clip[h]] : (long)clip[h] * p.T;
        if (rr >= Tc[h]) Tc[h] = 0;                                    // nothing of this half exists
    }
    // ---- activation tiles -> LDS (bf16): row lr of half h = clip row r0 - pad + lr, zero outside [0, T).  The loads of a batch of
    //      slots are all issued before the first conversion (unconditional, from a clamped row: a load under `if` is waited for on the
    //      spot, and 18 dependent global round trips per thread were a third of the kernel)
    {
        constexpr int C4 = CG / 4;
        constexpr int NSLOT = (2 * XROWS * C4 + PC_NT - 1) / PC_NT;
        constexpr int BATCH = 12;
#pragma unroll
        for (int s0 = 0; s0 < NSLOT; s0 += BATCH) {
            f32x4 v[BATCH];
            bool ok[BATCH];
            int dst[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int idx = tid + (s0 + q) * PC_NT;
                const bool in_tile = s0 + q < NSLOT && idx < 2 * XROWS * C4;
                const int ic = in_tile ? idx : 0;
                const int h = ic / (XROWS * C4), rem = ic - h * (XROWS * C4);
                const int lr = rem / C4, c4 = rem - lr * C4;
                const int cr = r0[h] - p.pad + lr;
                ok[q] = cr >= 0 && cr < Tc[h];
                dst[q] = in_tile ? h * XBYTES + lr * RS + c4 * 8 : -1;
                v[q] = *reinterpret_cast<const f32x4*>(p.x + (grow0[h] + (ok[q] ? cr : 0)) * p.H + g * CG + 4 * c4);
            }
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                if (dst[q] >= 0) *reinterpret_cast<bf16x4*>(Xs + dst[q]) = __builtin_convertvector(ok[q] ? v[q] : f32x4{0.f, 0.f, 0.f, 0.f}, bf16x4);
            }
        }
    }
    // ---- weight chunks: chunk c = flattened K elements [128 c, 128 c + 128) of every co: piece q of a thread = (co, 16-byte piece)
    const int nk = p.ntaps * CG / 32;                                  // k-steps
    const int nchunks = nk / PC_KC;
    const unsigned short* wg = p.w + (long)g * p.ntaps * p.Npad * CG;
    u32x4 rw[WSLOTS];
    auto issueW = [&](int c) {
#pragma unroll
        for (int q = 0; q < WSLOTS; ++q) {
            const int piece = tid + q * PC_NT;                         // co * (KC * 4) + j: j-th 8-element piece of the chunk
            const int co = piece / (PC_KC * 4), j = piece - co * (PC_KC * 4);
            const int f0 = c * (PC_KC * 32) + j * 8;                   // flattened (tap, ci) index: 8 elements inside one tap (CG % 8 == 0)
            const int tap = f0 / CG, ci = f0 - tap * CG;
            rw[q] = (piece < CG * PC_KC * 4) ? *reinterpret_cast<const u32x4*>(wg + ((long)tap * p.Npad + co) * CG + ci) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto storeW = [&](char* dst) {
#pragma unroll
        for (int q = 0; q < WSLOTS; ++q) {
            const int piece = tid + q * PC_NT;
            const int co = piece / (PC_KC * 4), j = piece - co * (PC_KC * 4);
            if (piece < CG * PC_KC * 4) *reinterpret_cast<u32x4*>(dst + co * WROW + j * 16) = rw[q];
        }
    };
    issueW(0);
    storeW(Ws);
    if (nchunks > 1) issueW(1);
    __syncthreads();

    // this wave's row tiles: 4 consecutive tiles of its half
    const int tile0 = (wave & 3) * 4;
    f32x4 acc[4][NTC];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xh = Xs + hsel * XBYTES + (tile0 * 16 + r16) * RS;
    // the lane's 8 elements of k-step ks start at flattened index 32 ks + 8 kg = (tap, ci); advanced by 32 per k-step
    int tap = (8 * kg) / CG, ci = 8 * kg - tap * CG;
    for (int c = 0; c < nchunks; ++c) {
        const char* wb = Ws + (c & 1) * WBYTES + r16 * WROW + kg * 16;
        // (fragment reads issued per k-step and scheduled by the compiler: an explicit one-step-ahead double buffer measured 94 us, all
        //  28 reads of a chunk up front 99 us, this form 85 us.  The loop is LDS-read-bound by its shape: N = Cg is three or four MFMA
        //  column tiles, so a wave reads 7 / 8 fragments per 12 / 16 MFMAs -- 146 B per cycle and CU at the matrix rate, above LDS's 128.)
#pragma unroll
        for (int ks = 0; ks < PC_KC; ++ks) {
            bf16x8 wf[NTC], xf[4];
#pragma unroll
            for (int j = 0; j < NTC; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wb + j * 16 * WROW + ks * 64);
            const char* xp = xh + tap * RS + ci * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xp + i * 16 * RS);
            // (every tile of the wave, also one past the clip's rows -- its LDS rows are zero: a per-tile `if` was compiled to exec-mask
            //  branches between groups of three MFMAs, with a wait in front of each)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NTC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            ci += 32;
            if (ci >= CG) { ci -= CG; ++tap; }
        }
        if (c + 1 < nchunks) {
            storeW(Ws + ((c + 1) & 1) * WBYTES);                       // (that buffer was last read during chunk c - 1: a barrier ago)
            if (c + 2 < nchunks) issueW(c + 2);
        }
        __syncthreads();
    }
    // ---- epilogue: lane (r16, kg) of tile (i, j): frame r0 + 16 (tile0 + i) + r16, channels g CG + 16 j + 4 kg ... + 3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = r0[hsel] + (tile0 + i) * 16 + r16;
        if (t < Tc[hsel]) {
            const long o = (grow0[hsel] + t) * p.H + g * CG + 4 * kg;
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + g * CG + 16 * j + 4 * kg);
                const f32x4 rv = *reinterpret_cast<const f32x4*>(p.x + o + 16 * j);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pc_gelu_erf(acc[i][j][e] + bv[e]) + rv[e];
                *reinterpret_cast<f32x4*>(p.out + o + 16 * j) = v;
            }
        }
    }
}

template <int CG>
static int posconv_launch(si_ctx* ctx, const PosConvParams& p, int nwg, double rows, hipStream_t st) {
    const size_t lds = 2 * (size_t)(PC_HALF_ROWS + 128) * (CG == 48 ? 96 : 160) + 2 * (size_t)CG * (PC_KC * 64 + 32);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(posconv_kernel<CG>), lds)) return rc;
    char name[32];
    snprintf(name, sizeof(name), "posconv_bf16_c%d", CG);
    si_prof_begin(ctx, name, 2.0 * rows * p.H * (double)CG * p.ntaps, rows * p.H * 12.0 + 2.0 * p.H * (double)CG * p.ntaps, st);
    hipLaunchKernelGGL(posconv_kernel<CG>, dim3(nwg, p.groups), dim3(PC_NT), lds, st, p);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller runs the tap-GEMM).
// x / out (rows, H) fp32; w = the tap-GEMM's packed bf16 weights W[g][tap][Npad][Cg]; row_off / lens (device, B) or null (uniform:
// clip b = rows [b T, (b + 1) T)); Tmax = the longest clip's frames (host); rows_total = the rows that exist (accounting).
int si_launch_posconv(si_ctx* ctx, const float* x, float* out, const void* w, const float* bias, int B, int T, int Tmax, int H, int groups,
                      int ntaps, int Npad, int pad, const int32_t* row_off, const int32_t* lens, double rows_total, hipStream_t st) {
    if (groups <= 0 || H % groups) return 1;
    const int CG = H / groups;
    if ((CG != 48 && CG != 64) || (ntaps * CG) % (32 * PC_KC) || pad < 0 || pad > 127 || ntaps > 128 || B <= 0 || Tmax <= 0) return 1;
    if ((reinterpret_cast<size_t>(x) & 15) || (reinterpret_cast<size_t>(out) & 15) || (reinterpret_cast<size_t>(w) & 15) || !bias) return 1;
    PosConvParams p{x, out, static_cast<const unsigned short*>(w), bias, row_off, lens, B, T, H, groups, ntaps, Npad, pad, 0, 1};
    p.pair = Tmax <= PC_HALF_ROWS ? 1 : 0;
    p.blocks_per_clip = (Tmax + 2 * PC_HALF_ROWS - 1) / (2 * PC_HALF_ROWS);
    const int nwg = p.pair ? (B + 1) / 2 : B * p.blocks_per_clip;
    return CG == 48 ? posconv_launch<48>(ctx, p, nwg, rows_total, st) : posconv_launch<64>(ctx, p, nwg, rows_total, st);
}
