// vocoder_kernels.hip -- the non-GEMM kernels of the HiFi-GAN half of the path (gfx950, wave64).
//  * extend_mel : time-only bilinear stretch x441/256, align_corners=False
//                 (I_ea/hifi_gan/inference_modified.py:16-19), fused with the channels-first -> channels-last
//                 transpose the generator kernels want; channel padding written as zeros.
//  * conv_post  : leaky_relu(0.01) -> Conv1d(C -> 1, k=7, pad 3) -> tanh (I_ea/hifi_gan/models.py:119-121),
//                 the tail of the generator: 1 output channel, so it is a per-sample dot product, not a GEMM.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void extend_mel_kernel(const float* __restrict__ mel, int D, int Tm, int Tout, int stretch,
                                                         float rscale, float* __restrict__ out, int ldo) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Tout * ldo) return;
    const int t = (int)(i / ldo), c = (int)(i - (long)t * ldo);
    float v = 0.f;
    if (c < D) {
        const float* row = mel + ((long)b * D + c) * Tm;
        if (stretch) {
            // ATen area_pixel_compute_source_index: one fused multiply-add in fp32, clamped at 0
            float src = fmaf((float)t + 0.5f, rscale, -0.5f);
            src = src < 0.f ? 0.f : src;
            int i0 = (int)floorf(src);
            if (i0 > Tm - 1) i0 = Tm - 1;
            const int i1 = i0 + 1 < Tm ? i0 + 1 : Tm - 1;
            float l1 = src - (float)i0;
            l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
            const float l0 = 1.f - l1;
            v = l0 * row[i0] + l1 * row[i1];
        } else {
            v = row[t];
        }
    }
    out[(long)b * Tout * ldo + i] = v;
}

int si_launch_extend_mel(si_ctx* ctx, const float* mel, int B, int D, int Tm, int Tout, int stretch, float* out, int ldo,
                         hipStream_t st) {
    if (B <= 0 || Tout <= 0) return SI_OK;
    const float rscale = (float)(1.0 / (441.0 / 256.0));
    dim3 grid((unsigned)(((long)Tout * ldo + 255) / 256), B);
    si_prof_begin(ctx, "extend_mel", 3.0 * B * Tout * D, 4.0 * B * D * ((double)Tm + Tout), st);
    hipLaunchKernelGGL(extend_mel_kernel, grid, dim3(256), 0, st, mel, D, Tm, Tout, stretch, rscale, out, ldo);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// 256 output samples per workgroup; the (256 + k - 1) x C input rows are staged (with the leaky-relu applied) into LDS as
// fp32 rows of C + 4 floats: 16-byte aligned, so a lane walks its rows with ds_read_b128, and for C = 32 the 36-dword
// stride puts the 16 lanes of every ds_read_b128 group on 16 different 4-bank slots (one lane = one output sample = one
// row per tap).  The kernel is bound by these reads (7 x 128 bytes per output sample): as 4-byte reads on a 33-float
// stride it took 114 us per 32 clips, three times the HBM time of its input.
// x16 (optional): the input as raw fp16 (the vocoder's fp16 activation stream) instead of fp32 x
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const unsigned short* __restrict__ x16,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int L, int C,
                                                        int k, float* __restrict__ wav) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ldx = C + 4;
    float* xs = reinterpret_cast<float*>(smem);              // [(256 + k - 1)][C + 4]
    float* ws = xs + (256 + k - 1) * ldx;                    // [k][C]
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 256;
    const int pad = k / 2;
    const int rows = 256 + k - 1;
    const int c4n = C / 4;
    const long xoff = (long)b * L * C;
    for (int idx = threadIdx.x; idx < rows * c4n; idx += 256) {
        const int r = idx / c4n, j = idx - r * c4n;
        const int t = t0 - pad + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < L) {
            if (x16) v = __builtin_convertvector(*reinterpret_cast<const f16x4*>(x16 + xoff + (long)t * C + 4 * j), f32x4);
            else v = *reinterpret_cast<const f32x4*>(x + xoff + (long)t * C + 4 * j);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
        *reinterpret_cast<f32x4*>(xs + r * ldx + 4 * j) = v;
    }
    for (int idx = threadIdx.x; idx < k * C; idx += 256) ws[idx] = w[idx];
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= L) return;
    float acc = bias[0];
    for (int kk = 0; kk < k; ++kk) {
        const float* xr = xs + (threadIdx.x + kk) * ldx;
        const float* wr = ws + kk * C;
        for (int c = 0; c < C; c += 4) {                     // the same fma order as a scalar walk over c
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c), wv = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(xv[e], wv[e], acc);
        }
    }
    wav[(long)b * L + t] = tanhf(acc);
}

int si_launch_conv_post(si_ctx* ctx, const float* x, const float* w, const float* bias, int B, int L, int C, int k, float* wav,
                        hipStream_t st, const unsigned short* x16) {
    if (C % 4 != 0) return si_fail(ctx, SI_EINVAL, "conv_post: C=%d must be a multiple of 4", C);
    if (B <= 0 || L <= 0) return SI_OK;
    const size_t lds = ((size_t)(256 + k - 1) * (C + 4) + (size_t)k * C) * sizeof(float);
    if (lds > 160 * 1024) return si_fail(ctx, SI_EINVAL, "conv_post: %d channels x %d taps need %zu bytes of LDS (> 160 KiB)", C, k, lds);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(conv_post_kernel), lds)) return rc;
    dim3 grid((L + 255) / 256, B);
    si_prof_begin(ctx, "conv_post", 2.0 * B * L * (double)C * k, (double)B * L * ((x16 ? 2.0 : 4.0) * C + 4.0), st);
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), lds, st, x, x16, w, bias, L, C, k, wav);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}


// ------------------------------------------------------------------------------------------------ f-2: CodeGenerator front
// I_da/src/model.py:148-189 (the LUT configuration, configs/*/hubert_lut.json): emb_c = emb_c_table[code] (:151-153),
// emb_p = emb_p_table[f0_code] (:160-161), the shorter of the two repeated frame-wise up to the longer (`_upsample`,
// :79-119: out[t] = in[t div (F / len)], lengths must divide), channel concat (:169), the speaker embedding VECTOR repeated
// over all frames and concatenated (:175-176; `emb_s = emb` at :157).  out (B, nparts * E, F) channels-first: the
// generator's input (si_hifigan_forward, stretch = 0).  One thread per output element; an index outside its table
// writes NaN (nn.Embedding would raise; a device kernel cannot).
__global__ __launch_bounds__(256) void unit_frontend_kernel(const int64_t* __restrict__ code, int Fc, const int64_t* __restrict__ f0c, int Fp,
                                                            const float* __restrict__ spk, const float* __restrict__ emb_c, int Kc,
                                                            const float* __restrict__ emb_p, int Kp, int E, int F, int nparts,
                                                            float* __restrict__ out) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= F) return;
    const int part = ch / E, e = ch - part * E;
    float v;
    if (part == 0) {
        const long idx = code[(long)b * Fc + t / (F / Fc)];
        v = (idx >= 0 && idx < Kc) ? emb_c[idx * E + e] : __builtin_nanf("");
    } else if (part == 1 && f0c) {
        const long idx = f0c[(long)b * Fp + t / (F / Fp)];
        v = (idx >= 0 && idx < Kp) ? emb_p[idx * E + e] : __builtin_nanf("");
    } else {
        v = spk[(long)b * E + e];
    }
    out[((long)b * nparts * E + ch) * F + t] = v;
}

int si_launch_unit_frontend(si_ctx* ctx, const int64_t* code, int Fc, const int64_t* f0_code, int Fp, const float* spk_emb,
                            const float* emb_c, int Kc, const float* emb_p, int Kp, int E, int B, float* out, hipStream_t st) {
    const int F = (f0_code && Fp > Fc) ? Fp : Fc;
    const int nparts = 1 + (f0_code ? 1 : 0) + (spk_emb ? 1 : 0);
    si_prof_begin(ctx, "unit_frontend", 0.0, 4.0 * B * nparts * E * (double)F, st);
    hipLaunchKernelGGL(unit_frontend_kernel, dim3((F + 255) / 256, nparts * E, B), dim3(256), 0, st, code, Fc, f0_code, Fp, spk_emb, emb_c, Kc,
                       emb_p, Kp, E, F, nparts, out);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ F0 VQ-VAE encoder (row f-2)
// One Conv1d of the Jukebox-style F0 encoder (I_da/src/modules/jukebox.py:11-116, resnet.py:29-97) on channels-first fp32
// (B, C, T): y[b][co][t] = bias[co] + sum_ci sum_k w[co][ci][k] * pre(x[b][ci][t * stride + k * dil - pad]) (+ res[b][co][t]),
// pre = ReLU or identity.  The network is tiny (width 32, 128 at the end; 800 -> 50 frames per 4 s clip): one thread
// per output element, weights out of L1/L2.  `cl` writes channels-last (B, T, C) -- the rows si_kmeans_assign reads.
__global__ __launch_bounds__(256) void small_conv1d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ res, float* __restrict__ y, int B, int Cin, int Tin,
                                                           int Cout, int Tout, int K, int stride, int dil, int pad, int relu_in, int cl) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * Cout * Tout;
    if (idx >= total) return;
    const int t = (int)(idx % Tout);
    const int co = (int)((idx / Tout) % Cout);
    const int b = (int)(idx / ((long)Tout * Cout));
    float acc = bias[co];
    const float* xb = x + (long)b * Cin * Tin;
    const float* wc = w + (long)co * Cin * K;
    for (int ci = 0; ci < Cin; ++ci) {
        for (int k = 0; k < K; ++k) {
            const int ti = t * stride + k * dil - pad;
            if (ti >= 0 && ti < Tin) {
                float v = xb[(long)ci * Tin + ti];
                if (relu_in) v = fmaxf(v, 0.f);
                acc = fmaf(wc[ci * K + k], v, acc);
            }
        }
    }
    if (res) acc += res[((long)b * Cout + co) * Tout + t];
    if (cl) y[((long)b * Tout + t) * Cout + co] = acc;
    else y[((long)b * Cout + co) * Tout + t] = acc;
}

int si_launch_small_conv1d(si_ctx* ctx, const float* x, const float* w, const float* bias, const float* res, float* y, int B, int Cin,
                           int Tin, int Cout, int Tout, int K, int stride, int dil, int pad, int relu_in, int channels_last, hipStream_t st) {
    const long total = (long)B * Cout * Tout;
    if (total <= 0) return SI_OK;
    si_prof_begin(ctx, "f0enc_conv1d", 2.0 * total * Cin * K, 4.0 * (total + (double)B * Cin * Tin), st);
    hipLaunchKernelGGL(small_conv1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, w, bias, res, y, B, Cin, Tin, Cout,
                       Tout, K, stride, dil, pad, relu_in, channels_last);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
