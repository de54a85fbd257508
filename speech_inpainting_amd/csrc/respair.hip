// respair.hip -- one ResBlock1 step  y' = y + conv2(lrelu(conv1(lrelu(y))))  as ONE kernel, for the narrow vocoder stages
// (C = 32 / 64 channels) in the fp16 activation-stream mode (gfx950, wave64, MFMA).
//
// I_ea/hifi_gan/models.py:36-43 runs, per (resblock, dilation): xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt);
// x = xt + x.  As two tap-GEMM launches the intermediate costs 4 of the pair's 10 bytes of HBM traffic per element, and
// these stages sit on the memory side.  With N = C <= 64 one workgroup owns ALL channels of its rows,
// so the intermediate can stay in LDS:
//   phase 1   t[R1 rows] = lrelu(conv1(lrelu(y)) + b1)      rows [m0 - p2, m0 - p2 + R1), zero outside the clip
//                                                           (conv2's zero padding applies to t), fp16 into LDS
//   phase 2   out[BMo rows] = (conv2(t) + b2 + y) * alpha (+ previous out)      BMo = R1 - (k - 1)
// The intermediate tile is written OVER the activation tile (dead once conv1 is done): 29.5 KB (C = 32) / 62 KB (C = 64)
// of LDS per workgroup, i.e. five / two resident workgroups per CU.
// conv1 is recomputed on the k - 1 halo rows between neighbouring tiles (1-8 %).  Weight slabs (C x C per tap) stream
// through a double buffer for both convolutions; activations arrive as raw fp16 (leaky-ReLU applied while staging), the
// residual is re-read from global (L2-hot) in the epilogue, which is row-contiguous (8-byte accesses after an LDS
// transpose, as tapgemm.hip's).  Arithmetic is that of the two-launch form in the same mode: fp16 operands, fp32
// accumulate, the intermediate rounded to fp16 once.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(256, 3) void respair_kernel(const ResPairParams p) {
    constexpr int R1 = 256;                              // intermediate rows per workgroup
    constexpr int TM = R1 / 128;                         // 32-row tiles per wave (4 waves)
    constexpr int TN = C / 32;
    constexpr int LD = C + 8;                            // LDS row stride in halves (16-byte pad)
    constexpr int KS = C / 16;                           // MFMA k-steps per tap
    constexpr int V8 = C / 8;                            // 16-byte vectors per weight row
    constexpr int WSLOTS = (C * V8 + 255) / 256;         // 16-byte weight vectors per thread and slab
    constexpr int V4 = C / 4;                            // 8-byte (4-half) slots per activation row
    constexpr int ASLOTS = C == 32 ? 10 : 20;            // (R1 + 50) * V4 / 256 rounded up

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int k = p.k, d = p.dil;
    const int p1 = d * (k - 1) / 2, p2 = (k - 1) / 2;
    const int BMo = R1 - (k - 1);
    const int R0 = R1 + (k - 1) * d;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * BMo;                     // first output row of this workgroup
    const int t_row0 = m0 - p2;                          // global row of intermediate row 0
    const int y_row0 = t_row0 - p1;                      // global row of staged activation row 0

    unsigned short* Ys = reinterpret_cast<unsigned short*>(smem);          // [R0][LD]   lrelu(y), fp16
    unsigned short* Ts = Ys;                                               // [R1][LD]   lrelu(conv1 + b1), fp16: OVER the activation
                                                                           // tile, which is dead once conv1 is done (one extra barrier)
    unsigned short* Ws = Ys + (size_t)(R1 + 50) * LD;                      // [2][C][LD] weight slab double buffer

    const long seg = (long)b * p.L * C;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.y16 + seg), 0, p.L * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out16 + seg, 0, p.L * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w1), 0, k * C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w2), 0, k * C * C * 2, 0x00020000);

    // ---- stage the activation tile: raw fp16 -> leaky-ReLU(0.1) on packed halves -> LDS (rows outside the clip read as zero)
    {
        const int r0 = tid / V4, j = tid - r0 * V4;
        f32x2 ra[ASLOTS];
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i)
            ra[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrsrc, ((y_row0 + r0 + i * (256 / V4)) * C + 4 * j) * 2, 0, 0));
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int r = r0 + i * (256 / V4);
            if (r < R0) {
                f16x4 h = __builtin_bit_cast(f16x4, ra[i]);
                const f16x4 hs = h * (_Float16)0.1f;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = h[e] > (_Float16)0 ? h[e] : hs[e];
                *reinterpret_cast<f16x4*>(Ys + (size_t)r * LD + 4 * j) = h;
            }
        }
    }
    // ---- weight slab streaming: slab s = tap s of conv1 for s < k, tap s - k of conv2 otherwise
    const int w_r0 = tid / V8, w_j = tid - w_r0 * V8;
    f32x4 rw[WSLOTS];
    auto issueW = [&](int s) {
        const bool second = s >= k;
        const int soff = (second ? s - k : s) * C * C * 2;
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) {
            const int r = w_r0 + i * (256 / V8);
            const int voff = r < C ? (r * C + 8 * w_j) * 2 : (int)0x80000000;
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? w2rsrc : w1rsrc, voff, soff, 0));
        }
    };
    auto storeW = [&](unsigned short* dst) {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) {
            const int r = w_r0 + i * (256 / V8);
            if (r < C) *reinterpret_cast<f32x4*>(dst + (size_t)r * LD + 8 * w_j) = rw[i];
        }
    };
    issueW(0);
    storeW(Ws);
    issueW(1);                                           // k >= 2: slab 1 exists
    __syncthreads();

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    const int wrow0 = wave * (R1 / 4);                   // this wave's first row of the R1-row tile
    // one tap: acc += A[rows wrow0 + .. (+ roff)] x W^T
    auto compute = [&](const unsigned short* As, int roff, const unsigned short* Wc) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 a[TM], w[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f16x8*>(As + (size_t)(wrow0 + i * 32 + l31 + roff) * LD + 16 * ks + 8 * half);
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const f16x8*>(Wc + (size_t)(j * 32 + l31) * LD + 16 * ks + 8 * half);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], w[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nslab = 2 * k;
    for (int s = 0; s < nslab; ++s) {
        const unsigned short* Wc = Ws + (size_t)(s & 1) * C * LD;
        if (s < k) {
            compute(Ys, s * d, Wc);                      // conv1 tap s: intermediate row r reads activation row r + s*d
        } else {
            compute(Ts, s - k, Wc);                      // conv2 tap: output row o reads intermediate row o + tap
        }
        if (s == k - 1) {
            __syncthreads();                             // every wave has finished reading the activation tile
            // ---- phase-1 epilogue: bias, leaky-ReLU, zero outside the clip, fp16, into the intermediate tile
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = j * 32 + l31;
                    const float bv = p.b1[n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int grow = t_row0 + row;
                        float v = acc[i][j][r] + bv;
                        v = v > 0.f ? v : 0.1f * v;
                        v = (grow >= 0 && grow < p.L) ? __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f) : 0.f;
                        Ts[(size_t)row * LD + n] = __builtin_bit_cast(unsigned short, (_Float16)v);
                    }
                }
            zero_acc();
        }
        if (s + 1 < nslab) {
            storeW(Ws + (size_t)((s + 1) & 1) * C * LD);  // slab s+1 (in flight since the previous iteration)
            if (s + 2 < nslab) issueW(s + 2);
            __syncthreads();
        }
    }

    // ---- final epilogue: 32x32 tiles transposed through a private LDS patch (the activation tile is dead), then
    //      row-contiguous 8-byte residual reads / stores
    __syncthreads();
    float* const tl = reinterpret_cast<float*>(smem) + wave * (32 * 36);
    const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = j * 32 + lc;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b2 + n);
#pragma unroll
            for (int r = 0; r < 16; ++r) tl[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = acc[i][j][r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = wrow0 + i * 32 + lr + 8 * q;                     // output row inside the tile
                const int grow = m0 + o;
                const int off = (o < BMo && grow < p.L) ? (grow * C + n) * 2 : (int)0x80000000;
                const f32x4 a = *reinterpret_cast<const f32x4*>(tl + (lr + 8 * q) * 36 + lc);
                const f32x4 res = __builtin_convertvector(__builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(yrsrc, off, 0, 0)), f32x4);
                f32x4 prev = {0.f, 0.f, 0.f, 0.f};
                if (p.accumulate) prev = __builtin_convertvector(__builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(orsrc, off, 0, 0)), f32x4);
                f16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = (a[e] + b4[e] + res[e]) * p.alpha + prev[e];
                    h[e] = (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), orsrc, off, 0, 0);
            }
        }
}

template <int C>
static int respair_launch(si_ctx* ctx, const ResPairParams& p, hipStream_t st) {
    constexpr int R1 = 256;
    constexpr int LD = C + 8;
    const int BMo = R1 - (p.k - 1);
    size_t lds = ((size_t)(R1 + 50) * LD + 2 * (size_t)C * LD) * 2;
    lds = std::max(lds, (size_t)4 * 32 * 36 * sizeof(float));
    auto kern = respair_kernel<C>;
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    char name[48];
    snprintf(name, sizeof(name), "respair_f16_c%d", C);
    const double elems = (double)p.B * p.L * C;
    si_prof_begin(ctx, name, 2.0 * 2.0 * elems * C * p.k, elems * (2.0 + 2.0 + 2.0 + (p.accumulate ? 2.0 : 0.0)) + 2.0 * 2.0 * p.k * C * C, st);
    hipLaunchKernelGGL(kern, dim3((p.L + BMo - 1) / BMo, p.B), dim3(256), lds, st, p);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller launches the two convolutions).
int si_launch_respair(si_ctx* ctx, int C, const unsigned short* y16, unsigned short* out16, const void* w1, const void* w2,
                      const float* b1, const float* b2, int B, int L, int k, int dil, float alpha, int accumulate, hipStream_t st) {
    if ((C != 32 && C != 64 && C != 128 && C != 256) || k < 3 || k > 11 || (k & 1) == 0 || (k - 1) * dil > 50 ||
        ((long)L + 512) * C * 2 >= (1L << 31)) return 1;
    if (!b1 || !b2) return 1;
    ResPairParams p{y16, out16, static_cast<const unsigned short*>(w1), static_cast<const unsigned short*>(w2), b1, b2, B, L, k, dil, alpha, accumulate};
    if (C >= 128) return si_launch_respair_wide(ctx, C, p, st);
    return C == 32 ? respair_launch<32>(ctx, p, st) : respair_launch<64>(ctx, p, st);
}
