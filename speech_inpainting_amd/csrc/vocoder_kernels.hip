// vocoder_kernels.hip -- the non-GEMM kernels of the HiFi-GAN half of the path (gfx950, wave64).
//  * extend_mel : time-only bilinear stretch x441/256, align_corners=False
//                 (I_ea/hifi_gan/inference_modified.py:16-19), fused with the channels-first -> channels-last
//                 transpose the generator kernels want; channel padding written as zeros.
//  * conv_post  : leaky_relu(0.01) -> Conv1d(C -> 1, k=7, pad 3) -> tanh (I_ea/hifi_gan/models.py:119-121),
//                 the tail of the generator: 1 output channel, so it is a per-sample dot product, not a GEMM.
#include <algorithm>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void extend_mel_kernel(const float* __restrict__ mel, int D, int Tms, int Touts, int stretch,
                                                         float rscale, float* __restrict__ out, int ldo,
                                                         const int32_t* __restrict__ tm_len, const int32_t* __restrict__ tout_len) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    // ragged batches: the clip's own frame counts inside rows of stride Tms / Touts (the interpolation clamps at ITS last frame)
    const int Tm = tm_len ? tm_len[b] : Tms, Tout = tout_len ? tout_len[b] : Touts;
    if (i >= (long)Tout * ldo) return;
    const int t = (int)(i / ldo), c = (int)(i - (long)t * ldo);
    float v = 0.f;
    if (c < D) {
        const float* row = mel + ((long)b * D + c) * Tms;
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
    out[(long)b * Touts * ldo + i] = v;
}

// The same stretch as a stand-alone step, channels-first in and out: (B, D, Tm) -> (B, D, Tout) -- the generator's own input layout,
// so that a WINDOW of the stretched mel can be vocoded (si_hifigan_forward*, stretch = 0).  The arithmetic is the statement above.
__global__ __launch_bounds__(256) void extend_mel_cf_kernel(const float* __restrict__ mel, int D, int Tm, int Tout, float rscale, float* __restrict__ out) {
    const int bc = blockIdx.y;                                         // b * D + c
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Tout) return;
    const float* row = mel + (long)bc * Tm;
    float src = fmaf((float)t + 0.5f, rscale, -0.5f);
    src = src < 0.f ? 0.f : src;
    int i0 = (int)floorf(src);
    if (i0 > Tm - 1) i0 = Tm - 1;
    const int i1 = i0 + 1 < Tm ? i0 + 1 : Tm - 1;
    float l1 = src - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    const float l0 = 1.f - l1;
    out[(long)bc * Tout + t] = l0 * row[i0] + l1 * row[i1];
}

int si_launch_extend_mel_cf(si_ctx* ctx, const float* mel, int B, int D, int Tm, int Tout, float* out, hipStream_t st) {
    if (B <= 0 || Tout <= 0) return SI_OK;
    const float rscale = (float)(1.0 / (441.0 / 256.0));
    si_prof_begin(ctx, "extend_mel", 3.0 * B * Tout * D, 4.0 * B * D * ((double)Tm + Tout), st);
    hipLaunchKernelGGL(extend_mel_cf_kernel, dim3((Tout + 255) / 256, B * D), dim3(256), 0, st, mel, D, Tm, Tout, rscale, out);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_extend_mel(si_ctx* ctx, const float* mel, int B, int D, int Tm, int Tout, int stretch, float* out, int ldo,
                         hipStream_t st, const int32_t* tm_len, const int32_t* tout_len) {
    if (B <= 0 || Tout <= 0) return SI_OK;
    const float rscale = (float)(1.0 / (441.0 / 256.0));
    dim3 grid((unsigned)(((long)Tout * ldo + 255) / 256), B);
    si_prof_begin(ctx, "extend_mel", 3.0 * B * Tout * D, 4.0 * B * D * ((double)Tm + Tout), st);
    hipLaunchKernelGGL(extend_mel_kernel, grid, dim3(256), 0, st, mel, D, Tm, Tout, stretch, rscale, out, ldo, tm_len, tout_len);
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
                                                        const float* __restrict__ w, const float* __restrict__ bias, int Ls, int C,
                                                        int k, float* __restrict__ wav, const int32_t* __restrict__ lens) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ldx = C + 4;
    float* xs = reinterpret_cast<float*>(smem);              // [(256 + k - 1)][C + 4]
    float* ws = xs + (256 + k - 1) * ldx;                    // [k][C]
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 256;
    const int L = lens ? lens[b] : Ls;                        // ragged batches: the clip's own rows at stride Ls
    if (t0 >= L) return;
    const int pad = k / 2;
    const int rows = 256 + k - 1;
    const int c4n = C / 4;
    const long xoff = (long)b * Ls * C;
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
    wav[(long)b * Ls + t] = tanhf(acc);
}

// The same tail on the fp16 activation stream (C = 32), on the matrix pipe: out[t] = sum_tap w[tap] . y[t + tap] is an N = 1 GEMM;
// as 16-column MFMAs with 15 zero columns it wastes 15/16 of the pipe and still takes a tenth of the time the per-lane dot
// product spends on its LDS reads (7 x 8 ds_read_b128 of activations + as many of weights per output sample; here 7 fragment
// reads per 16 samples, the weights in registers).  512 samples per workgroup (33 KB of LDS: four workgroups per CU overlap each
// other's staging): the (512 + 6) x 32 rows are staged with the
// leaky-ReLU(0.01) applied to the packed halves (max(x, 0.01 x)), 64-byte rows in reschain.hip's swizzle; orientation
// D^T = W . Y^T: lane (time row, k group 0) holds output channel 0 in accumulator element 0.  Operands are fp16 (the mode's
// rounding everywhere else: activations are stored that way, the 224 weights are rounded once), accumulation fp32.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
#define CPM_ROWS 512
template <bool VL>
__global__ __launch_bounds__(256) void conv_post_mfma_kernel(const unsigned short* __restrict__ x16, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B, int L, float* __restrict__ wav,
                                                             const int32_t* __restrict__ lens, int total_tiles) {
    constexpr int C = 32, K = 7, PAD = 3, NR = CPM_ROWS + K - 1;
    __shared__ __attribute__((aligned(16))) char ys[(NR + 2) * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    // weights of tap k as the A operand: row n = output channel (only n = 0 is real), this lane's 8 channels 8 kg ... 8 kg + 7
    f16x8 wf[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[k][e] = r16 == 0 ? (_Float16)w[k * C + 8 * kg + e] : (_Float16)0.f;
    const float b0 = bias[0];
    // persistent workgroups: tile = (clip, 512-sample block); the rows of the NEXT tile are requested before this one is
    // computed (all of them in flight at once: one HBM round trip per tile, hidden behind the previous tile's work)
    const int tiles_x = (L + CPM_ROWS - 1) / CPM_ROWS, total = VL ? total_tiles : tiles_x * B;
    auto tile_of = [&](int t, int& tb, int& tt0, int& tL) {            // tile -> clip, first sample, the clip's samples (ragged: si_vl_tile)
        if constexpr (VL) { const SiVlTile v = si_vl_tile(lens, B, CPM_ROWS, t); tb = v.b; tt0 = v.row0; tL = v.L; }
        else { tb = t / tiles_x; tt0 = (t - tb * tiles_x) * CPM_ROWS; tL = L; }
    };
    constexpr int NSLOT = (NR * 4 + 255) / 256;
    u32x4v raw[NSLOT];
    auto issue = [&](int tile) {
        int tb, tt0, tL;
        tile_of(tile, tb, tt0, tL);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x16 + (long)tb * L * C), 0, tL * C * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int q = tid + i * 256;
            const int t = tt0 - PAD + (q >> 2);                        // rows outside the clip read as zero through the descriptor
            raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t < 0 || q >= NR * 4) ? (int)0x80000000 : (t * C + 8 * (q & 3)) * 2, 0, 0);
        }
    };
    issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        int b, t0, Lb;
        tile_of(tile, b, t0, Lb);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int q = tid + i * 256;
            const int r = q >> 2, ch = q & 3;
            f16x8 h = __builtin_bit_cast(f16x8, raw[i]);
            h = __builtin_elementwise_max(h, h * (_Float16)0.01f);    // leaky_relu, default slope (models.py:119)
            if (q < NR * 4) *reinterpret_cast<f16x8*>(ys + r * 64 + ((ch << 4) ^ (((r >> 1) & 3) << 4))) = h;
        }
        __syncthreads();
        issue(tile + (int)gridDim.x < total ? tile + (int)gridDim.x : tile);   // clamped: unconditional loads
        // wave w: row tiles w, w + 4, ...; a tap moves the fragment by one row, so the swizzled address is formed per (tile, tap)
        for (int rt = wave; rt < CPM_ROWS / 16; rt += 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int lin0 = (rt * 16 + r16) * 64 + (kg << 4);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int lin = lin0 + k * 64;
                const f16x8 y = *reinterpret_cast<const f16x8*>(ys + (lin ^ ((lin >> 3) & 0x30)));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[k], y, acc, 0, 0, 0);
            }
            const int t = t0 + rt * 16 + r16;
            if (kg == 0 && t < Lb) wav[(long)b * L + t] = tanhf(acc[0] + b0);
        }
        __syncthreads();                                               // the tile is consumed: the next one may be staged
    }
}

int si_launch_conv_post(si_ctx* ctx, const float* x, const float* w, const float* bias, int B, int L, int C, int k, float* wav,
                        hipStream_t st, const unsigned short* x16, const int32_t* lens, const int32_t* lens_host) {
    if (C % 4 != 0) return si_fail(ctx, SI_EINVAL, "conv_post: C=%d must be a multiple of 4", C);
    if (B <= 0 || L <= 0) return SI_OK;
    if ((lens == nullptr) != (lens_host == nullptr)) return si_fail(ctx, SI_EINVAL, "conv_post: ragged batches need the lengths on the device and on the host");
    double rows = (double)B * L;
    if (lens_host) { rows = 0; for (int b = 0; b < B; ++b) rows += lens_host[b]; }
    if (x16 && C == 32 && k == 7 && (long)L * C * 2 < (1L << 31)) {    // the fp16 stream's tail on the matrix pipe
        const int total = lens ? (int)si_vl_tiles(lens_host, B, CPM_ROWS) : ((L + CPM_ROWS - 1) / CPM_ROWS) * B;
        if (total <= 0) return SI_OK;
        si_prof_begin(ctx, "conv_post", 2.0 * rows * (double)C * k, rows * (2.0 * C + 4.0), st);
        if (lens) hipLaunchKernelGGL(conv_post_mfma_kernel<true>, dim3(std::min(total, si_num_cus(ctx) * 4)), dim3(256), 0, st, x16, w, bias, B, L, wav, lens, total);
        else hipLaunchKernelGGL(conv_post_mfma_kernel<false>, dim3(std::min(total, si_num_cus(ctx) * 4)), dim3(256), 0, st, x16, w, bias, B, L, wav, lens, total);
        si_prof_end(ctx, st);
        SI_HIP_CHECK(hipGetLastError());
        return SI_OK;
    }
    const size_t lds = ((size_t)(256 + k - 1) * (C + 4) + (size_t)k * C) * sizeof(float);
    if (lds > 160 * 1024) return si_fail(ctx, SI_EINVAL, "conv_post: %d channels x %d taps need %zu bytes of LDS (> 160 KiB)", C, k, lds);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(conv_post_kernel), lds)) return rc;
    dim3 grid((L + 255) / 256, B);
    si_prof_begin(ctx, "conv_post", 2.0 * rows * (double)C * k, rows * ((x16 ? 2.0 : 4.0) * C + 4.0), st);
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), lds, st, x, x16, w, bias, L, C, k, wav, lens);
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


// The whole F0 encoder as ONE launch (row f-2; I_da/src/modules/jukebox.py:11-116, resnet.py:29-97): one workgroup per F0 track keeps
// every activation of that track in LDS (width 32: 800 -> 400 -> ... -> 50 frames: two buffers of at most T / 2 x 32 floats = 51 KB --
// a res block's 1 x 1 convolution writes x + conv(relu(t)) IN PLACE over x: each output reads only its own x) and walks the 37
// convolutions with workgroup barriers between them.  As 37 launches of a one-thread-per-output kernel the encoder took 0.78 ms per
// call for 0.4 GFLOP (21 us per launch, all of it latency).  Per layer the weights are staged into LDS transposed to [ci][tap][co], so
// that a thread that owns 8 output channels of one frame reads them as two 16-byte broadcasts per input value: 3 LDS reads per 8 FMAs.
// Every output is the same fmaf chain as small_conv1d_kernel's (bias first, then ci outer / tap inner, padding taps skipped):
// bit-identical results.
struct F0EncParams {
    const float* weights; const float* f0; float* h_out;
    int in_width, out_width, width, n_state, depth, down_t, stride_t, growth, T;
    int dk, dpad;                     // kernel / padding of the strided convolutions
    int slot;                         // floats per LDS activation buffer
    int wmax;                         // floats of the largest layer's weights + bias
};

// global w[co][ci][k], bias[co] -> LDS wl[ci][k][co], bl[co]
__device__ __forceinline__ void f0_stage_w(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wl, float* __restrict__ bl,
                                           int Cout, int Cin, int K) {
    const int n = Cout * Cin * K;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int co = i / (Cin * K), r = i - co * (Cin * K);           // r = ci * K + k
        wl[r * Cout + co] = w[i];
    }
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) bl[i] = bias[i];
}

// one convolution out of LDS weights: a thread owns 8 consecutive output channels of one output frame.  x: [ci][x_ct] (LDS, or
// global for the first layer); y: LDS [co][Tout] (may alias res when K == 1), or global channels-last when to_global_cl.
__device__ __forceinline__ void f0_conv8(const float* __restrict__ x, int x_ct, const float* __restrict__ wl, const float* __restrict__ bl,
                                         const float* res, float* y, int Cin, int Tin, int Cout, int Tout, int K, int stride, int dil, int pad,
                                         bool relu_in, bool to_global_cl) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int items = (Cout / 8) * Tout;
    for (int item = threadIdx.x; item < items; item += blockDim.x) {
        const int t = item % Tout, co0 = (item / Tout) * 8;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = bl[co0 + c];
        // (a tap in the padding contributes fmaf(w, 0, acc) = acc: the same value as skipping it, without a branch around the loads,
        //  so that the compiler can keep several (ci, tap) steps of LDS reads in flight)
        const int t_in0 = t * stride - pad;
#pragma unroll 4
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xr = x + (long)ci * x_ct;
            for (int k = 0; k < K; ++k) {
                const int ti = t_in0 + k * dil;
                const bool in = ti >= 0 && ti < Tin;
                float v = xr[in ? ti : 0];
                if (relu_in) v = fmaxf(v, 0.f);
                v = in ? v : 0.f;
                const float* wp = wl + (ci * K + k) * Cout + co0;
                const f32x4v w0 = *reinterpret_cast<const f32x4v*>(wp), w1 = *reinterpret_cast<const f32x4v*>(wp + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) { acc[c] = fmaf(w0[c], v, acc[c]); acc[4 + c] = fmaf(w1[c], v, acc[4 + c]); }
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a = acc[c];
            if (res) a += res[(long)(co0 + c) * Tout + t];
            if (to_global_cl) y[(long)t * Cout + co0 + c] = a;           // channels-last rows for the bottleneck's arg-min
            else y[(long)(co0 + c) * Tout + t] = a;
        }
    }
}

__global__ __launch_bounds__(512) void f0enc_fused_kernel(const F0EncParams p) {
    extern __shared__ __attribute__((aligned(16))) char f0_smem[];
    float* X = reinterpret_cast<float*>(f0_smem);
    float* Y = X + p.slot;
    float* wl = Y + p.slot;                                            // [ci][k][co] of the current layer
    float* bl = wl + p.wmax;
    const int b = blockIdx.x;
    const float* w = p.weights;
    const float* x = p.f0 + (long)b * p.in_width * p.T;               // (in_width, T) in global memory
    int cin = p.in_width, Tc = p.T;
    for (int i = 0; i < p.down_t; ++i) {
        const int To = (Tc + 2 * p.dpad - p.dk) / p.stride_t + 1;
        const float* cw = w; w += (long)p.width * cin * p.dk;
        const float* cb = w; w += p.width;
        f0_stage_w(cw, cb, wl, bl, p.width, cin, p.dk);
        __syncthreads();
        f0_conv8(x, Tc, wl, bl, nullptr, Y, cin, Tc, p.width, To, p.dk, p.stride_t, 1, p.dpad, false, false);
        __syncthreads();
        { float* t_ = X; X = Y; Y = t_; }                              // X = this block's residual stream, Y = scratch
        x = X; cin = p.width; Tc = To;
        int dil = 1;
        for (int j = 0; j < p.depth; ++j) {
            const float* w3 = w; w += (long)p.n_state * p.width * 3;
            const float* b3 = w; w += p.n_state;
            const float* w1 = w; w += (long)p.width * p.n_state;
            const float* b1 = w; w += p.width;
            f0_stage_w(w3, b3, wl, bl, p.n_state, p.width, 3);
            __syncthreads();
            f0_conv8(X, Tc, wl, bl, nullptr, Y, p.width, Tc, p.n_state, Tc, 3, 1, dil, dil, true, false);
            __syncthreads();
            f0_stage_w(w1, b1, wl, bl, p.width, p.n_state, 1);
            __syncthreads();
            f0_conv8(Y, Tc, wl, bl, X, X, p.n_state, Tc, p.width, Tc, 1, 1, 1, 0, true, false);   // in place: x <- x + conv1(relu(t))
            __syncthreads();
            dil *= p.growth;
        }
    }
    const float* fw = w; w += (long)p.out_width * p.width * 3;
    f0_stage_w(fw, w, wl, bl, p.out_width, p.width, 3);
    __syncthreads();
    f0_conv8(X, Tc, wl, bl, nullptr, p.h_out + (long)b * Tc * p.out_width, p.width, Tc, p.out_width, Tc, 3, 1, 1, 1, false, true);
}

// SI_OK when launched, negative on error, 1 when the track does not fit LDS (the caller launches the convolutions one by one)
int si_launch_f0enc_fused(si_ctx* ctx, const float* weights, const float* f0, int B, int T, float* h_out, int in_width, int out_width, int width,
                          int n_state, int depth, int down_t, int stride_t, int growth, int dk, int dpad, double macs, hipStream_t st) {
    const int T1 = (T + 2 * dpad - dk) / stride_t + 1;                 // the longest intermediate
    const size_t slot = ((size_t)std::max(width, n_state) * std::max(T1, 1) + 3) / 4 * 4;
    size_t wmax = std::max((size_t)width * std::max(in_width, width) * dk, std::max((size_t)n_state * width * 3, (size_t)out_width * width * 3));
    wmax = (wmax + 3) / 4 * 4;
    const size_t lds = (2 * slot + wmax + std::max(std::max(width, n_state), out_width)) * sizeof(float);
    if (lds > 158 * 1024 || B <= 0 || width % 8 || n_state % 8 || out_width % 8) return 1;
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(f0enc_fused_kernel), lds)) return rc;
    F0EncParams p{weights, f0, h_out, in_width, out_width, width, n_state, depth, down_t, stride_t, growth, T, dk, dpad, (int)slot, (int)wmax};
    si_prof_begin(ctx, "f0enc_fused", 2.0 * macs, 4.0 * B * ((double)in_width * T + 0.0), st);
    hipLaunchKernelGGL(f0enc_fused_kernel, dim3(B), dim3(512), lds, st, p);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
