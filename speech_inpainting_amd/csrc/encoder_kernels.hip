// encoder_kernels.hip -- the non-GEMM kernels of the HuBERT half of the path (gfx950, wave64).
//
//  * wave_stats / conv0 family : zero-mask + zero-mean/unit-variance normalise (I_ea/predict.py:132-141)
//    fused into the load of conv0 -> GroupNorm(C groups) -> GELU (modeling_hubert.py:154-175).
//    GroupNorm over time needs per-(clip, channel) mean/variance of the conv output.  Because conv0 has ONE
//    input channel, those are quadratic forms of the clip's K-lag autocorrelation:
//        sum_t y_c[t]   = sum_k w[c][k] * S[k],          S[k]    = sum_t x[s*t+k]
//        sum_t y_c[t]^2 = sum_kk' w[c][k] w[c][k'] R[kk'], R[kk'] = sum_t x[s*t+k] x[s*t+k']
//    so the statistics cost one pass over the 256 KB clip (in fp64) instead of a pass over the 26 MB conv
//    output, and the big tensor is written exactly once, already normalised and activated (HBM-bound row A1).
//  * layernorm : one wave per row, two-pass in registers, wave shuffles for the reductions.
//  * attention : flash-style exact-fp32 attention on v_mfma_f32_32x32x2_f32 with the "keys on the accumulator
//    rows" orientation: S^T = K Q^T leaves each lane holding one query's scores, so the softmax needs one
//    cross-half exchange, and the S^T accumulator registers ARE the B operand of O^T = V^T P^T (no LDS
//    round trip, no shuffles; cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's operand").
//  * codebook_splice : cosine arg-max against centred centroids + raw-centroid splice (I_ea/loss_fn.py:44-47,
//    I_ea/predict.py:164-168,184-187).
#include <algorithm>
#include <cstdlib>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One raw sample as the normalisation sees it: [+ pre_add in fp64, rounded once], then the zero mask.
__device__ __forceinline__ float masked_sample(float v, int i, int ms, int ml, bool has_add, double add) {
    if (i >= ms && i < ms + ml) return 0.f;
    return has_add ? (float)((double)v + add) : v;
}

// ------------------------------------------------------------------------------------------------ A0
// stats[b] = {mean, 1/sqrt(var + eps)} of the zero-masked clip, accumulated in fp64.
__global__ __launch_bounds__(1024) void wave_stats_kernel(WaveNormParams p, double* __restrict__ stats) {
    const int b = blockIdx.x;
    if (!p.normalize) {
        if (threadIdx.x == 0) { stats[2 * b] = 0.0; stats[2 * b + 1] = 1.0; }
        return;
    }
    const float* x = p.wav + (long)b * p.N;
    const int ms = p.mask_start ? p.mask_start[b] : 0;
    const int ml = p.mask_len ? p.mask_len[b] : 0;
    const int nv = p.valid_len ? min(max(p.valid_len[b], 1), p.N) : p.N;      // real samples of a right-padded clip
    const bool has_add = p.pre_add != nullptr;
    const double add = has_add ? p.pre_add[b] : 0.0;
    double s = 0.0, ss = 0.0;
    // 16 bytes per lane and four loads in flight per thread (one workgroup walks a whole clip: with 4-byte loads in a
    // dependent loop the launch was a 35 us latency chain); the tail and unaligned clips take the scalar loop
    const bool vec = (p.N & 3) == 0 && (reinterpret_cast<size_t>(p.wav) & 15) == 0;
    const int nv4 = vec ? nv & ~3 : 0;
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < nv4; i += blockDim.x * 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(x + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = masked_sample(q[e], i + e, ms, ml, has_add, add);
            s += v;
            ss += (double)v * v;
        }
    }
    for (int i = nv4 + threadIdx.x; i < nv; i += blockDim.x) {
        const float v = masked_sample(x[i], i, ms, ml, has_add, add);
        s += v;
        ss += (double)v * v;
    }
    __shared__ double red[2][16];
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = s; red[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0, SS = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { S += red[0][i]; SS += red[1][i]; }
        const double mean = S / nv;
        double var = SS / nv - mean * mean;
        if (var < 0) var = 0;
        stats[2 * b] = mean;
        stats[2 * b + 1] = 1.0 / sqrt(var + (double)p.norm_eps);
    }
}

__device__ __forceinline__ float load_norm(const float* x, int i, int ms, int ml, float mean, float rstd, int nv, bool has_add, double add) {
    if (i >= nv) return 0.f;                                         // padding value, applied after the normalisation
    return (masked_sample(x[i], i, ms, ml, has_add, add) - mean) * rstd;
}

// partials[b][chunk][NP]: NP = K + K(K+1)/2 lag sums of the normalised clip over the chunk's conv positions.
#define SI_C0_TCH 512
__global__ __launch_bounds__(256) void conv0_lagsums_kernel(WaveNormParams p, const double* __restrict__ stats,
                                                            double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                       // S*TCH + K
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int nchunks = gridDim.x;
    const int t0 = chunk * SI_C0_TCH;
    const int L1 = p.seg_L1 ? p.seg_L1[b] : p.L1;                     // ragged batches: the clip's own conv0 rows
    const int nt = min(SI_C0_TCH, L1 - t0);
    const int K = p.K, S = p.S;
    const int NP = K + K * (K + 1) / 2;
    if (nt <= 0) {                                                    // a chunk past the clip: exact zeros keep the fixed-order sum
        if ((int)threadIdx.x < NP) partials[((long)b * nchunks + chunk) * NP + threadIdx.x] = 0.0;   // equal to the clip's own
        return;
    }
    const float mean = (float)stats[2 * b], rstd = (float)stats[2 * b + 1];
    const int ms = p.mask_start ? p.mask_start[b] : 0, ml = p.mask_len ? p.mask_len[b] : 0;
    const float* x = p.wav + (long)b * p.N;
    const int win = (nt - 1) * S + K;
    const int nv = p.valid_len ? min(max(p.valid_len[b], 1), p.N) : p.N;
    const bool has_add = p.pre_add != nullptr;
    const double add = has_add ? p.pre_add[b] : 0.0;
    for (int i = threadIdx.x; i < win; i += 256) xs[i] = load_norm(x, t0 * S + i, ms, ml, mean, rstd, nv, has_add, add);
    __syncthreads();
    __shared__ double part[2][160];
    const int pr = threadIdx.x & 127, half = threadIdx.x >> 7;
    double acc = 0.0;
    if (pr < NP) {
        int k0, k1;
        if (pr < K) { k0 = pr; k1 = -1; }
        else {                                  // pair index -> (k0 <= k1)
            int q = pr - K; k0 = 0;
            while (q >= K - k0) { q -= K - k0; ++k0; }
            k1 = k0 + q;
        }
        for (int t = half; t < nt; t += 2) {
            const float a = xs[t * S + k0];
            acc += (k1 < 0) ? (double)a : (double)a * (double)xs[t * S + k1];
        }
        part[half][pr] = acc;
    }
    __syncthreads();
    if (half == 0 && pr < NP) partials[((long)b * nchunks + chunk) * NP + pr] = part[0][pr] + part[1][pr];
}

// affine[b][c] = {a, sh}:  y = gelu(a * conv + sh)  with a = gamma*rstd_c, sh = beta - mean_c*a
__global__ __launch_bounds__(256) void conv0_gn_affine_kernel(WaveNormParams p, const double* __restrict__ partials, int nchunks,
                                                              const float* __restrict__ w, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ affine) {
    __shared__ double tot[160];
    const int b = blockIdx.x;
    const int K = p.K, NP = K + K * (K + 1) / 2;
    if ((int)threadIdx.x < NP) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += partials[((long)b * nchunks + c) * NP + threadIdx.x];   // fixed order
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) {
        const float* wc = w + (long)c * K;
        double m = 0.0, e2 = 0.0;
        for (int k = 0; k < K; ++k) m += (double)wc[k] * tot[k];
        int q = K;
        for (int k0 = 0; k0 < K; ++k0)
            for (int k1 = k0; k1 < K; ++k1, ++q)
                e2 += (k0 == k1 ? 1.0 : 2.0) * (double)wc[k0] * (double)wc[k1] * tot[q];
        const int L1 = p.seg_L1 ? p.seg_L1[b] : p.L1;
        m /= L1;
        double var = e2 / L1 - m * m;
        if (var < 0) var = 0;
        const double rstd = 1.0 / sqrt(var + 1e-5);
        const double a = (double)gamma[c] * rstd;
        affine[((long)b * p.C + c) * 2] = (float)a;
        affine[((long)b * p.C + c) * 2 + 1] = (float)((double)beta[c] - m * a);
    }
}

__global__ void conv0_bias_affine_kernel(int B, int C, const float* __restrict__ bias, float* __restrict__ affine) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * C) { affine[2 * i] = 1.f; affine[2 * i + 1] = bias ? bias[i % C] : 0.f; }
}

// out[b][t][c] = act(a[b][c] * sum_k w[c][k] * xhat[b][s*t+k] + sh[b][c]); channels-last, written once.
// FAST (the bf16-only output of the bf16 encoder mode): erf by the GEMM epilogues' 15-operation form (si_gelu_fast, |error| <= 1.5e-7
// before the value is rounded to 8 bits): 152 -> 140 us for 32 clips.
#define SI_C0_ROWS 64
template <int K, bool GELU, bool FAST = false>
__global__ __launch_bounds__(256) void conv0_apply_kernel(WaveNormParams p, const double* __restrict__ stats,
                                                          const float* __restrict__ w, const float* __restrict__ affine,
                                                          float* __restrict__ out, unsigned short* __restrict__ out16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * SI_C0_ROWS;
    const int nt = min(SI_C0_ROWS, (p.seg_L1 ? p.seg_L1[b] : p.L1) - t0);
    if (nt <= 0) return;                                              // ragged batches: rows past the clip's own are not written
    const int S = p.S, C = p.C;
    const float mean = (float)stats[2 * b], rstd = (float)stats[2 * b + 1];
    const int ms = p.mask_start ? p.mask_start[b] : 0, ml = p.mask_len ? p.mask_len[b] : 0;
    const float* x = p.wav + (long)b * p.N;
    const int win = (nt - 1) * S + K;
    const int nv = p.valid_len ? min(max(p.valid_len[b], 1), p.N) : p.N;
    const bool has_add = p.pre_add != nullptr;
    const double add = has_add ? p.pre_add[b] : 0.0;
    for (int i = threadIdx.x; i < win; i += 256) xs[i] = load_norm(x, t0 * S + i, ms, ml, mean, rstd, nv, has_add, add);
    __syncthreads();
    const int tpr = C / 4;                       // threads per output row (float4 of channels each)
    const int rpp = 256 / tpr;                   // rows per pass
    const int c4 = (threadIdx.x % tpr) * 4;
    const int r0 = threadIdx.x / tpr;
    float wr[4][K];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < K; ++k) wr[e][k] = w[(long)(c4 + e) * K + k];
    float av[4], sv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        av[e] = affine[((long)b * C + c4 + e) * 2];
        sv[e] = affine[((long)b * C + c4 + e) * 2 + 1];
    }
    for (int r = r0; r < nt; r += rpp) {
        float xv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) xv[k] = xs[r * S + k];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) acc = fmaf(wr[e][k], xv[k], acc);
            float v = fmaf(av[e], acc, sv[e]);
            y[e] = GELU ? (FAST ? si_gelu_fast(v) : gelu_erf(v)) : v;
        }
        // out16: the only consumer is a bf16 GEMM-form conv -- write its operand (half the 26 MB/clip) instead of fp32
        // (16-byte stores of eight channels per thread, and the taps and the erf on packed fp32 operations: the same 140 us --
        //  the kernel runs at the rate of its 419 MB of stores, 2.9 TB/s)
        if (out16) *reinterpret_cast<bf16x4*>(out16 + ((long)b * p.L1 + t0 + r) * C + c4) = __builtin_convertvector(y, bf16x4);
        else *reinterpret_cast<f32x4*>(out + ((long)b * p.L1 + t0 + r) * C + c4) = y;
    }
}

size_t si_conv0_partials_bytes(int B, int N) {
    // upper bound on chunks: L1 <= N
    const long nchunks = ((long)N + SI_C0_TCH - 1) / SI_C0_TCH + 1;
    return (size_t)B * nchunks * 160 * sizeof(double);
}

static int conv0_check(si_ctx* ctx, const WaveNormParams& p) {
    if (p.K != 10) return si_fail(ctx, SI_EINVAL, "conv0 kernel size %d unsupported (HuBERT uses 10)", p.K);
    const int tpr = p.C / 4;
    if (p.C % 4 != 0 || tpr > 256 || (256 % tpr) != 0)
        return si_fail(ctx, SI_EINVAL, "conv0 channel count %d must be 4*2^j <= 1024", p.C);
    return SI_OK;
}

int si_launch_wave_stats(si_ctx* ctx, const WaveNormParams& p, double* stats, hipStream_t st) {
    si_prof_begin(ctx, "wave_stats", 3.0 * p.B * p.N, 4.0 * p.B * p.N, st);
    hipLaunchKernelGGL(wave_stats_kernel, dim3(p.B), dim3(1024), 0, st, p, stats);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

static int conv0_apply(si_ctx* ctx, const WaveNormParams& p, const double* stats, const float* w, const float* affine,
                       float* out, bool gelu, hipStream_t st, unsigned short* out16 = nullptr) {
    dim3 grid((p.L1 + SI_C0_ROWS - 1) / SI_C0_ROWS, p.B);
    const size_t lds = ((size_t)(SI_C0_ROWS - 1) * p.S + p.K) * sizeof(float);
    si_prof_begin(ctx, "conv0_apply", 2.0 * p.B * p.L1 * (double)p.C * p.K, p.B * (4.0 * p.N + (out16 ? 2.0 : 4.0) * p.L1 * p.C), st);
    if (gelu && out16) hipLaunchKernelGGL((conv0_apply_kernel<10, true, true>), grid, dim3(256), lds, st, p, stats, w, affine, out, out16);
    else if (gelu) hipLaunchKernelGGL((conv0_apply_kernel<10, true>), grid, dim3(256), lds, st, p, stats, w, affine, out, out16);
    else hipLaunchKernelGGL((conv0_apply_kernel<10, false>), grid, dim3(256), lds, st, p, stats, w, affine, out, out16);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_conv0_groupnorm(si_ctx* ctx, const WaveNormParams& p, const double* stats, const float* w, const float* gamma,
                              const float* beta, double* partials, float* affine, float* out, hipStream_t st,
                              unsigned short* out16) {
    int rc = conv0_check(ctx, p);
    if (rc) return rc;
    const int nchunks = (p.L1 + SI_C0_TCH - 1) / SI_C0_TCH;
    const size_t lds = ((size_t)(SI_C0_TCH - 1) * p.S + p.K) * sizeof(float);
    si_prof_begin(ctx, "conv0_lagsums", 2.0 * p.B * p.L1 * 65.0, 4.0 * p.B * p.N, st);
    hipLaunchKernelGGL(conv0_lagsums_kernel, dim3(nchunks, p.B), dim3(256), lds, st, p, stats, partials);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    si_prof_begin(ctx, "conv0_gn_affine", 2.0 * p.B * p.C * 65.0, 8.0 * p.B * p.C, st);
    hipLaunchKernelGGL(conv0_gn_affine_kernel, dim3(p.B), dim3(256), 0, st, p, partials, nchunks, w, gamma, beta, affine);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return conv0_apply(ctx, p, stats, w, affine, out, true, st, out16);
}

// plain flavour with an explicit affine scratch (used by the layer-norm feature extractor)
int si_launch_conv0_affine(si_ctx* ctx, const WaveNormParams& p, const double* stats, const float* w, const float* bias,
                           float* affine, float* out, hipStream_t st) {
    int rc = conv0_check(ctx, p);
    if (rc) return rc;
    const int n = p.B * p.C;
    hipLaunchKernelGGL(conv0_bias_affine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.B, p.C, bias, affine);
    SI_HIP_CHECK(hipGetLastError());
    return conv0_apply(ctx, p, stats, w, affine, out, false, st);
}

// ------------------------------------------------------------------------------------------------ padded batches
struct ConvStack { int n; int k[SI_MAX_CONV], s[SI_MAX_CONV]; };
__global__ void frame_lengths_kernel(const int32_t* __restrict__ valid_len, int B, ConvStack cs, int T, int32_t* __restrict__ valid_frames) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = valid_len[b];
    for (int i = 0; i < cs.n; ++i) n = n >= cs.k[i] ? (n - cs.k[i]) / cs.s[i] + 1 : 0;      // floor((n - k) / s) + 1
    valid_frames[b] = min(max(n, 1), T);
}
int si_launch_frame_lengths(si_ctx* ctx, const int32_t* valid_len, int B, int nconv, const int32_t* kernels, const int32_t* strides,
                            int T, int32_t* valid_frames, hipStream_t st) {
    ConvStack cs{};
    cs.n = nconv;
    for (int i = 0; i < nconv; ++i) { cs.k[i] = kernels[i]; cs.s[i] = strides[i]; }
    hipLaunchKernelGGL(frame_lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, st, valid_len, B, cs, T, valid_frames);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
__global__ __launch_bounds__(256) void zero_padded_rows_kernel(float* __restrict__ x, int T, int H, const int32_t* __restrict__ valid_frames) {
    const int b = blockIdx.y, t = blockIdx.x;
    if (t < valid_frames[b]) return;
    float* r = x + ((long)b * T + t) * H;
    for (int i = threadIdx.x * 4; i < H; i += 1024) *reinterpret_cast<f32x4*>(r + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}
int si_launch_zero_padded_rows(si_ctx* ctx, float* x, int B, int T, int H, const int32_t* valid_frames, hipStream_t st) {
    if (H % 4) return si_fail(ctx, SI_EINVAL, "zero_padded_rows: width %d must be a multiple of 4", H);
    hipLaunchKernelGGL(zero_padded_rows_kernel, dim3(T, B), dim3(256), 0, st, x, T, H, valid_frames);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ragged batches: rows of clip b, [0, T_b) of a (B, Tmax, C) padded tensor <-> rows [row_off[b], row_off[b + 1]) of a packed one.
// UNPACK also zeroes the padded rows [T_b, Tmax) of the destination (the output is defined on all of it).
template <bool UNPACK>
__global__ __launch_bounds__(256) void repack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int Tmax, int C,
                                                          const int32_t* __restrict__ row_off) {
    const int b = blockIdx.y;
    const int r0 = row_off[b], Tb = row_off[b + 1] - r0;
    const int c4n = C / 4;
    for (int t = blockIdx.x; t < (UNPACK ? Tmax : Tb); t += gridDim.x) {
        const long pad = ((long)b * Tmax + t) * C, pk = ((long)r0 + t) * C;
        for (int i = threadIdx.x; i < c4n; i += 256) {
            if (UNPACK) *reinterpret_cast<f32x4*>(dst + pad + 4 * i) = t < Tb ? *reinterpret_cast<const f32x4*>(src + pk + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            else *reinterpret_cast<f32x4*>(dst + pk + 4 * i) = *reinterpret_cast<const f32x4*>(src + pad + 4 * i);
        }
    }
}
int si_launch_repack_rows(si_ctx* ctx, const float* src, float* dst, int B, int Tmax, int C, const int32_t* row_off, bool unpack, hipStream_t st) {
    if (C % 4) return si_fail(ctx, SI_EINVAL, "repack_rows: width %d must be a multiple of 4", C);
    if (B <= 0 || Tmax <= 0) return SI_OK;
    dim3 grid((unsigned)std::min(Tmax, 512), B);
    si_prof_begin(ctx, unpack ? "unpack_rows" : "pack_rows", 0.0, 8.0 * B * Tmax * C, st);
    if (unpack) hipLaunchKernelGGL(repack_rows_kernel<true>, grid, dim3(256), 0, st, src, dst, Tmax, C, row_off);
    else hipLaunchKernelGGL(repack_rows_kernel<false>, grid, dim3(256), 0, st, src, dst, Tmax, C, row_off);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row at a time, C % 4 == 0, C <= 2048 (the row lives in 8 float4 registers per lane).  Waves are
// persistent: wave w walks rows w, w + W, ... and requests its next row before it reduces the current one, so the reads
// of one row overlap the two reductions and the stores of the previous one (one row per wave and launch had every wave
// of the chip load, then reduce, then store in step: 16 us for 49 MB).
template <bool GELU, int NS>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, unsigned short* __restrict__ y16, long rows, int C,
                                                        float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 nx[NS];
    auto fetch = [&](long r) {
        const float* xr = x + r * C;
        const float* ar = add ? add + r * C : nullptr;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = (j * 64 + lane) * 4;
            nx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < C) {
                nx[j] = *reinterpret_cast<const f32x4*>(xr + i);
                if (ar) nx[j] += *reinterpret_cast<const f32x4*>(ar + i);
            }
        }
    };
    fetch(row);
    for (; row < rows; row += nwaves) {
        f32x4 v[NS];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NS; ++j) { v[j] = nx[j]; s += v[j][0] + v[j][1] + v[j][2] + v[j][3]; }
        if (row + nwaves < rows) fetch(row + nwaves);
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = (j * 64 + lane) * 4;
            if (i < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + eps);
        if (stats && lane == 0) *reinterpret_cast<f32x2*>(stats + 2 * row) = f32x2{mean, rstd};   // for a consumer that recomputes y (si_ln_apply)
        float* yr = y + row * C;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = (j * 64 + lane) * 4;
            if (i < C) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + i);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = si_ln_apply(v[j][e], mean, rstd, g[e], bt[e]);
                    o[e] = GELU ? (y ? gelu_erf(t) : si_gelu_fast(t)) : t;   // bf16-only output: the GEMM epilogues' erf (common.h)
                }
                if (y) *reinterpret_cast<f32x4*>(yr + i) = o;           // (y16 alone: the only consumer is a bf16 GEMM)
                // operand-ready copy for a bf16 GEMM consumer (round-to-nearest-even, as the GEMM's own staging would)
                if (y16) *reinterpret_cast<bf16x4*>(y16 + row * C + i) = __builtin_convertvector(o, bf16x4);
            }
        }
    }
}

int si_launch_layernorm(si_ctx* ctx, const float* x, const float* add, const float* gamma, const float* beta, float* y,
                        long rows, int C, float eps, int gelu, hipStream_t st, unsigned short* y16, float* stats) {
    if (C % 4 != 0 || C > 2048) return si_fail(ctx, SI_EINVAL, "layernorm width %d must be a multiple of 4 and <= 2048", C);
    if (rows <= 0) return SI_OK;
    // 16 waves per CU (four 4-wave workgroups), each walking its share of the rows (same-box A/B over 8 / 16 / 32 waves
    // per CU and one row per wave: 0.40 / 0.35 / 0.36 / 0.37 ms per step)
    // C <= 1024 (every width on the path): the row lives in 4 float4 registers per lane instead of 8 -- 56 VGPRs instead of 122, eight
    // waves per SIMD instead of four -- and 32 waves per CU give every row of the B*T = 6368 launches its own wave: one HBM round
    // trip for the whole launch instead of one and a half.
    const bool narrow = C <= 1024;
    const int per_cu = narrow ? 8 : 4;
    dim3 grid((unsigned)std::min<long>((rows + 3) / 4, (long)si_num_cus(ctx) * per_cu));
    if (!y && !y16) return si_fail(ctx, SI_EINVAL, "layernorm: no output");
    si_prof_begin(ctx, "layernorm", 8.0 * rows * C, (4.0 + (add ? 4.0 : 0.0) + (y ? 4.0 : 0.0) + (y16 ? 2.0 : 0.0)) * rows * C, st);
    if (gelu && narrow) hipLaunchKernelGGL((layernorm_kernel<true, 4>), grid, dim3(256), 0, st, x, add, gamma, beta, y, y16, rows, C, eps, stats);
    else if (gelu) hipLaunchKernelGGL((layernorm_kernel<true, 8>), grid, dim3(256), 0, st, x, add, gamma, beta, y, y16, rows, C, eps, stats);
    else if (narrow) hipLaunchKernelGGL((layernorm_kernel<false, 4>), grid, dim3(256), 0, st, x, add, gamma, beta, y, y16, rows, C, eps, stats);
    else hipLaunchKernelGGL((layernorm_kernel<false, 8>), grid, dim3(256), 0, st, x, add, gamma, beta, y, y16, rows, C, eps, stats);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ attention
// head_dim = 64.  Workgroup = 4 waves = 128 queries of one (clip, head); wave = 32 queries.  Keys/values stream
// through LDS in tiles of 32.  Per tile and wave:
//   S^T[key][q] = sum_d K[key][d] * Q[q][d]/8      A = K tile rows (LDS, b128), B = Q row kept in 32 registers
//   online softmax down each lane's column (query = lane&31): in-lane over 16 registers + one lane^32 exchange
//   O^T[d][q] += sum_key V[key][d] * P^T[key][q]   A = V tile (LDS, b32), B = the S^T accumulator registers:
//     MFMA step s reads key (s&3)+8*(s>>2)+4*(lane>>5) from both operands, which is exactly the key whose score
//     register s of this lane holds (C/D row map of the 32x32 MFMA).
#define ATT_KT 32
#define ATT_LD 68     // 64 + 4 floats: 16-B aligned rows, consecutive rows shift by 4 banks

// row_off (B + 1) or null: ragged batches, PACKED rows -- clip b is rows [row_off[b], row_off[b + 1]) of qkv / out and its
// frame count is their difference (instead of b * T ... and T); the grid covers the longest clip, blocks past a clip exit.
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        unsigned short* __restrict__ out16, int T, int H,
                                                        int heads, const int32_t* __restrict__ valid_frames,
                                                        const int32_t* __restrict__ row_off) {
    __shared__ __attribute__((aligned(16))) float Ks[ATT_KT * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_KT * ATT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const long ld = 3L * H;
    const long row0 = row_off ? row_off[b] : (long)b * T;
    if (row_off) { T = row_off[b + 1] - row_off[b]; if ((int)blockIdx.x * 128 >= T) return; }
    const float* base = qkv + row0 * ld + h * 64;
    const int Tk = valid_frames ? min(max(valid_frames[b], 1), T) : T;   // keys beyond it are padding (every query row still runs)

    // this lane's half of its query row, pre-scaled by head_dim^-0.5 = 2^-3 (exact)
    float qr[32];
    {
        const int qrow = min(q0 + l31, T - 1);
        const float* qp = base + (long)qrow * ld + half * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) qr[4 * j + e] = t[e] * 0.125f;
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;

    for (int k0 = 0; k0 < Tk; k0 += ATT_KT) {
        __syncthreads();
        // stage K and V tiles: 32 rows x 64 floats each; 256 threads x 2 float4 per matrix
        for (int idx = tid; idx < ATT_KT * 16; idx += 256) {
            const int r = idx >> 4, j = idx & 15;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + r < T) {
                const float* rp = base + (long)(k0 + r) * ld;
                kv = *reinterpret_cast<const f32x4*>(rp + H + 4 * j);
                vv = *reinterpret_cast<const f32x4*>(rp + 2 * H + 4 * j);
            }
            *reinterpret_cast<f32x4*>(Ks + r * ATT_LD + 4 * j) = kv;
            *reinterpret_cast<f32x4*>(Vs + r * ATT_LD + 4 * j) = vv;
        }
        __syncthreads();
        // S^T tile
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Ks + l31 * ATT_LD + half * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(kp + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qr[4 * j + e], s, 0, 0, 0);
        }
        // mask keys beyond T, column max
        float tmax = -INFINITY;
        if (k0 + ATT_KT > Tk) {                                        // only the last key tile has keys to mask (uniform branch: the
#pragma unroll                                                         // compare / select per score was a third of the loop's VALU work)
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= Tk) s[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);              // finite: every tile holds at least key k0 < T
        const float alpha = __expf(mrun - mnew);           // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mnew); psum += s[r]; }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // O^T += V^T P^T
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int krow = (st & 3) + 8 * (st >> 2) + 4 * half;
            const float v0 = Vs[krow * ATT_LD + l31];
            const float v1 = Vs[krow * ATT_LD + 32 + l31];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[st], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[st], o1, 0, 0, 0);
        }
    }
    // O^T[d][q]: column = lane&31 = query, rows d = (r&3) + 8*(r>>2) + 4*half (+32 for o1)
    const int q = q0 + l31;
    if (q < T) {
        const float inv = 1.0f / lrun;
        const long o = (row0 + q) * H + h * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 8 * g4 + 4 * half;
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            if (out16) {                 // operand-ready bf16 for the output projection (the only consumer)
                *reinterpret_cast<bf16x4*>(out16 + o + d) = __builtin_convertvector(a, bf16x4);
                *reinterpret_cast<bf16x4*>(out16 + o + 32 + d) = __builtin_convertvector(c, bf16x4);
            } else {
                *reinterpret_cast<f32x4*>(out + o + d) = a;
                *reinterpret_cast<f32x4*>(out + o + 32 + d) = c;
            }
        }
    }
}

// bf16 MFMA form for the bf16 encoder mode: same tiling and online softmax (fp32), but S^T = K Q^T and O^T = V^T P^T on
// v_mfma_f32_32x32x16_bf16 (4 + 4 instructions per 32-key tile instead of 32 + 32 fp32 ones).  K is staged as bf16
// [key][d]; V is staged TRANSPOSED as bf16 [d][key], because the P^T operand comes straight out of the score
// accumulators: k-slot i of lane-half h of MFMA step st is score register 8*st + i, i.e. key 16*st + 4*h + (i & 3) +
// 8*(i >> 2) -- two runs of four consecutive keys, which the V^T operand reads as two 8-byte LDS loads of row d.
#define ATB_LDK 72    // 64 + 8 halves: 144-byte rows
#define ATB_LDV 36    // 32 + 4 halves: 72-byte rows = 18 banks: the 32 rows of a ds_read_b64 group land on 32 different bank pairs (80-byte rows: rows r and r + 16 collide; PMC conflict share of the kernel 55 %)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void attention_bf16_kernel(const float* __restrict__ qkv, unsigned short* __restrict__ out16, int T,
                                                             int H, int heads, const int32_t* __restrict__ valid_frames,
                                                             const int32_t* __restrict__ row_off) {
    __shared__ __attribute__((aligned(16))) unsigned short Ks[ATT_KT * ATB_LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[64 * ATB_LDV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const long ld = 3L * H;
    const long row0 = row_off ? row_off[b] : (long)b * T;
    if (row_off) { T = row_off[b + 1] - row_off[b]; if ((int)blockIdx.x * 128 >= T) return; }
    const float* base = qkv + row0 * ld + h * 64;
    const int Tk = valid_frames ? min(max(valid_frames[b], 1), T) : T;

    bf16x8 qb[4];                                          // query row, 8 dims per k-step and lane half, scaled by 2^-3
    {
        const int qrow = min(q0 + l31, T - 1);
        const float* qp = base + (long)qrow * ld;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 8 * half);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 8 * half + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qb[ks][e] = (__bf16)(t0[e] * 0.125f); qb[ks][4 + e] = (__bf16)(t1[e] * 0.125f); }
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;

    for (int k0 = 0; k0 < Tk; k0 += ATT_KT) {
        __syncthreads();
        for (int idx = tid; idx < ATT_KT * 16; idx += 256) {
            const int r = idx >> 4, j = idx & 15;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + r < T) kv = *reinterpret_cast<const f32x4*>(base + (long)(k0 + r) * ld + H + 4 * j);
            *reinterpret_cast<bf16x4*>(Ks + r * ATB_LDK + 4 * j) = __builtin_convertvector(kv, bf16x4);
        }
        // V transposed: a thread takes two consecutive keys x four dims and writes four packed 32-bit words
        {
            const int rp = tid >> 4, j = tid & 15;                  // key pair 0..15, dim group 0..15
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (k0 + 2 * rp < T) v0 = *reinterpret_cast<const f32x4*>(base + (long)(k0 + 2 * rp) * ld + 2 * H + 4 * j);
            if (k0 + 2 * rp + 1 < T) v1 = *reinterpret_cast<const f32x4*>(base + (long)(k0 + 2 * rp + 1) * ld + 2 * H + 4 * j);
            // (whole-vector bit casts: extracting single __bf16 lanes with __builtin_bit_cast returned lane 0 for every index)
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t w0 = __builtin_bit_cast(u32x2_t, __builtin_convertvector(v0, bf16x4));
            const u32x2_t w1 = __builtin_bit_cast(u32x2_t, __builtin_convertvector(v1, bf16x4));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = (w0[e >> 1] >> (16 * (e & 1))) & 0xffffu, hi = (w1[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                *reinterpret_cast<unsigned*>(Vt + (4 * j + e) * ATB_LDV + 2 * rp) = lo | (hi << 16);
            }
        }
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + l31 * ATB_LDK + 16 * ks + 8 * half);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[ks], s, 0, 0, 0);
        }
        float tmax = -INFINITY;
        if (k0 + ATT_KT > Tk) {                                        // only the last key tile has keys to mask (uniform branch: the
#pragma unroll                                                         // compare / select per score was a third of the loop's VALU work)
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= Tk) s[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mnew); psum += s[r]; }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 pb;
#pragma unroll
            for (int i = 0; i < 8; ++i) pb[i] = (__bf16)s[8 * st + i];
            const unsigned short* v0p = Vt + l31 * ATB_LDV + 16 * st + 4 * half;
            const unsigned short* v1p = Vt + (32 + l31) * ATB_LDV + 16 * st + 4 * half;
            bf16x8 va, vc;
            const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(v0p), a1 = *reinterpret_cast<const bf16x4*>(v0p + 8);
            const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(v1p), c1 = *reinterpret_cast<const bf16x4*>(v1p + 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = a0[e]; va[4 + e] = a1[e]; vc[e] = c0[e]; vc[4 + e] = c1[e]; }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc, pb, o1, 0, 0, 0);
        }
    }
    const int q = q0 + l31;
    if (q < T) {
        const float inv = 1.0f / lrun;
        const long o = (row0 + q) * H + h * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 8 * g4 + 4 * half;
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<bf16x4*>(out16 + o + d) = __builtin_convertvector(a, bf16x4);
            *reinterpret_cast<bf16x4*>(out16 + o + 32 + d) = __builtin_convertvector(c, bf16x4);
        }
    }
}

// The same kernel on a bf16 q | k | v matrix (written as such by the QKV GEMM's epilogue: the rounding this kernel's staging
// applied to the fp32 matrix, moved into the producer -- bit-identical scores, half the bytes written and read).  The
// 2^-3 query scale is exact in bf16.
__global__ __launch_bounds__(256) void attention_bf16in_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out16, int T,
                                                               int H, int heads, const int32_t* __restrict__ valid_frames,
                                                               const int32_t* __restrict__ row_off) {
    __shared__ __attribute__((aligned(16))) unsigned short Ks[ATT_KT * ATB_LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[64 * ATB_LDV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const long ld = 3L * H;
    const long row0 = row_off ? row_off[b] : (long)b * T;
    if (row_off) { T = row_off[b + 1] - row_off[b]; if ((int)blockIdx.x * 128 >= T) return; }
    const unsigned short* base = qkv + row0 * ld + h * 64;
    const int Tk = valid_frames ? min(max(valid_frames[b], 1), T) : T;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

    bf16x8 qb[4];
    {
        const int qrow = min(q0 + l31, T - 1);
        const unsigned short* qp = base + (long)qrow * ld;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 t = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(qp + 16 * ks + 8 * half));
#pragma unroll
            for (int e = 0; e < 8; ++e) qb[ks][e] = (__bf16)((float)t[e] * 0.125f);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;

    // The next key tile is requested while the current one is computed on (one tile = 4 + 4 MFMAs per wave: a load issued
    // and consumed inside the same trip left the whole L2 round trip exposed, seven times per launch).  Rows past the
    // clip are clamped to its last row: they are masked below (key >= Tk) or belong to query rows that are not stored.
    const int kr = tid >> 3, kj = tid & 7;                             // K tile: 32 rows x 128 bytes, one 16-byte chunk per thread
    const int vrp = tid >> 4, vj = tid & 15;                           // V: two consecutive keys x four dims per thread
    u32x4_t kv;
    u32x2_t w0, w1;
    auto fetch = [&](int k0) {
        kv = *reinterpret_cast<const u32x4_t*>(base + (long)min(k0 + kr, T - 1) * ld + H + 8 * kj);
        w0 = *reinterpret_cast<const u32x2_t*>(base + (long)min(k0 + 2 * vrp, T - 1) * ld + 2 * H + 4 * vj);
        w1 = *reinterpret_cast<const u32x2_t*>(base + (long)min(k0 + 2 * vrp + 1, T - 1) * ld + 2 * H + 4 * vj);
    };
    fetch(0);
    for (int k0 = 0; k0 < Tk; k0 += ATT_KT) {
        __syncthreads();
        *reinterpret_cast<u32x4_t*>(Ks + kr * ATB_LDK + 8 * kj) = kv;
        {   // V transposed: four packed 32-bit words (key pair) per thread
            const bool in0 = k0 + 2 * vrp < T, in1 = k0 + 2 * vrp + 1 < T;     // zero, not a clamped copy: P * V must not see NaN payloads of stale rows
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = in0 ? (w0[e >> 1] >> (16 * (e & 1))) & 0xffffu : 0u, hi = in1 ? (w1[e >> 1] >> (16 * (e & 1))) & 0xffffu : 0u;
                *reinterpret_cast<unsigned*>(Vt + (4 * vj + e) * ATB_LDV + 2 * vrp) = lo | (hi << 16);
            }
        }
        __syncthreads();
        fetch(k0 + ATT_KT < Tk ? k0 + ATT_KT : k0);                   // unconditional (clamped): a conditional load drains vmcnt at the join
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + l31 * ATB_LDK + 16 * ks + 8 * half);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[ks], s, 0, 0, 0);
        }
        float tmax = -INFINITY;
        if (k0 + ATT_KT > Tk) {                                        // only the last key tile has keys to mask (uniform branch: the
#pragma unroll                                                         // compare / select per score was a third of the loop's VALU work)
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= Tk) s[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mnew); psum += s[r]; }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 pb;
#pragma unroll
            for (int i = 0; i < 8; ++i) pb[i] = (__bf16)s[8 * st + i];
            const unsigned short* v0p = Vt + l31 * ATB_LDV + 16 * st + 4 * half;
            const unsigned short* v1p = Vt + (32 + l31) * ATB_LDV + 16 * st + 4 * half;
            bf16x8 va, vc;
            const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(v0p), a1 = *reinterpret_cast<const bf16x4*>(v0p + 8);
            const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(v1p), c1 = *reinterpret_cast<const bf16x4*>(v1p + 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = a0[e]; va[4 + e] = a1[e]; vc[e] = c0[e]; vc[4 + e] = c1[e]; }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc, pb, o1, 0, 0, 0);
        }
    }
    const int q = q0 + l31;
    if (q < T) {
        const float inv = 1.0f / lrun;
        const long o = (row0 + q) * H + h * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 8 * g4 + 4 * half;
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<bf16x4*>(out16 + o + d) = __builtin_convertvector(a, bf16x4);
            *reinterpret_cast<bf16x4*>(out16 + o + 32 + d) = __builtin_convertvector(c, bf16x4);
        }
    }
}

// T <= 256 (every 4 s clip: T = 199): ONE workgroup per (clip, head) with one wave per 32 queries and the clip's WHOLE K and V
// staged once -- one barrier per launch instead of two per key tile, K / V fetched once per (clip, head) instead of once per
// 128 queries, and no query block that is mostly padding (T = 199 as 128 + 71).  Per key tile the arithmetic is the tiled
// kernel's, statement for statement: bit-identical output.  LDS: per key tile a [32][ATB_LDK] block of K and a [64][ATB_LDV]
// block of V^T (the tiled kernel's layouts): 9 216 bytes per tile, 64.5 KB at T = 199 -- two workgroups per CU.
__global__ __launch_bounds__(512) void attention_bf16in_whole_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out16, int T,
                                                                     int H, int heads, const int32_t* __restrict__ valid_frames,
                                                                     const int32_t* __restrict__ row_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned short att_smem[];
    const int nkt = (int)(blockDim.x >> 6);                            // key tiles == waves == ceil(T / 32)
    unsigned short* const Ks = att_smem;                               // [nkt][32][ATB_LDK]
    unsigned short* const Vt = att_smem + nkt * ATT_KT * ATB_LDK;      // [nkt][64][ATB_LDV]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int q0 = wave * 32;
    const long ld = 3L * H;
    const long row0 = row_off ? row_off[b] : (long)b * T;
    if (row_off) T = row_off[b + 1] - row_off[b];                      // (every wave stays: they all stage K / V; waves past the clip store nothing)
    const unsigned short* base = qkv + row0 * ld + h * 64;
    const int Tk = valid_frames ? min(max(valid_frames[b], 1), T) : T;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

    // ---- K and V of the clip -> LDS: nkt * 256 slots (one 16-byte chunk of K and one key pair x four dims of V each) over
    //      64 * nkt threads: four slots per thread, all loads in flight before the first store
    u32x4_t kv[4];
    u32x2_t w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + i * (int)blockDim.x;
        const int k0 = (s >> 8) * ATT_KT, in = s & 255;
        const int kr = in >> 3, kj = in & 7, vrp = in >> 4, vj = in & 15;
        kv[i] = *reinterpret_cast<const u32x4_t*>(base + (long)min(k0 + kr, T - 1) * ld + H + 8 * kj);
        w0[i] = *reinterpret_cast<const u32x2_t*>(base + (long)min(k0 + 2 * vrp, T - 1) * ld + 2 * H + 4 * vj);
        w1[i] = *reinterpret_cast<const u32x2_t*>(base + (long)min(k0 + 2 * vrp + 1, T - 1) * ld + 2 * H + 4 * vj);
    }
    bf16x8 qb[4];
    {
        const int qrow = min(q0 + l31, T - 1);
        const unsigned short* qp = base + (long)qrow * ld;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 t = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(qp + 16 * ks + 8 * half));
#pragma unroll
            for (int e = 0; e < 8; ++e) qb[ks][e] = (__bf16)((float)t[e] * 0.125f);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + i * (int)blockDim.x;
        const int kt = s >> 8, k0 = kt * ATT_KT, in = s & 255;
        const int kr = in >> 3, kj = in & 7, vrp = in >> 4, vj = in & 15;
        *reinterpret_cast<u32x4_t*>(Ks + (kt * ATT_KT + kr) * ATB_LDK + 8 * kj) = kv[i];
        const bool in0 = k0 + 2 * vrp < T, in1 = k0 + 2 * vrp + 1 < T;         // zero, not a clamped copy (see the tiled kernel)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = in0 ? (w0[i][e >> 1] >> (16 * (e & 1))) & 0xffffu : 0u, hi = in1 ? (w1[i][e >> 1] >> (16 * (e & 1))) & 0xffffu : 0u;
            *reinterpret_cast<unsigned*>(Vt + (kt * 64 + 4 * vj + e) * ATB_LDV + 2 * vrp) = lo | (hi << 16);
        }
    }
    __syncthreads();

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -INFINITY, lrun = 0.f;
    for (int k0 = 0; k0 < Tk; k0 += ATT_KT) {
        const unsigned short* Kt = Ks + k0 * ATB_LDK;
        const unsigned short* Vtt = Vt + (k0 / ATT_KT) * 64 * ATB_LDV;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kt + l31 * ATB_LDK + 16 * ks + 8 * half);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[ks], s, 0, 0, 0);
        }
        float tmax = -INFINITY;
        if (k0 + ATT_KT > Tk) {                                        // only the last key tile has keys to mask (uniform branch: the
#pragma unroll                                                         // compare / select per score was a third of the loop's VALU work)
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= Tk) s[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mnew); psum += s[r]; }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 pb;
#pragma unroll
            for (int i = 0; i < 8; ++i) pb[i] = (__bf16)s[8 * st + i];
            const unsigned short* v0p = Vtt + l31 * ATB_LDV + 16 * st + 4 * half;
            const unsigned short* v1p = Vtt + (32 + l31) * ATB_LDV + 16 * st + 4 * half;
            bf16x8 va, vc;
            const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(v0p), a1 = *reinterpret_cast<const bf16x4*>(v0p + 8);
            const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(v1p), c1 = *reinterpret_cast<const bf16x4*>(v1p + 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = a0[e]; va[4 + e] = a1[e]; vc[e] = c0[e]; vc[4 + e] = c1[e]; }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc, pb, o1, 0, 0, 0);
        }
    }
    const int q = q0 + l31;
    if (q < T) {
        const float inv = 1.0f / lrun;
        const long o = (row0 + q) * H + h * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 8 * g4 + 4 * half;
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<bf16x4*>(out16 + o + d) = __builtin_convertvector(a, bf16x4);
            *reinterpret_cast<bf16x4*>(out16 + o + 32 + d) = __builtin_convertvector(c, bf16x4);
        }
    }
}

int si_launch_attention_bf16in(si_ctx* ctx, const unsigned short* qkv16, int B, int T, int H, int heads, hipStream_t st, unsigned short* out16,
                               const int32_t* valid_frames, const int32_t* row_off, double real_t2) {
    if (heads <= 0 || H != heads * 64) return si_fail(ctx, SI_EINVAL, "attention kernel needs head_dim 64 (H=%d heads=%d)", H, heads);
    if (B <= 0 || T <= 0) return SI_OK;
    dim3 grid((T + 127) / 128, B * heads);
    const int nkt = (T + ATT_KT - 1) / ATT_KT;
    const size_t lds = (size_t)nkt * (ATT_KT * ATB_LDK + 64 * ATB_LDV) * sizeof(unsigned short);
    if (T <= 256) if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(attention_bf16in_whole_kernel), lds)) return rc;   // (before the profile bracket opens)
    // (ragged batches: T = the longest clip; real_t2 = sum of T_b^2 for the algorithmic count)
    si_prof_begin(ctx, "attention_bf16", 4.0 * (real_t2 > 0 ? real_t2 : B * (double)T * T) * H, 8.0 * B * T * H, st);
    if (T <= 256) {                                                    // the whole clip's K / V in LDS, one workgroup per (clip, head)
        hipLaunchKernelGGL(attention_bf16in_whole_kernel, dim3(B * heads), dim3(64 * nkt), lds, st, qkv16, out16, T, H, heads, valid_frames, row_off);
    } else
    hipLaunchKernelGGL(attention_bf16in_kernel, grid, dim3(256), 0, st, qkv16, out16, T, H, heads, valid_frames, row_off);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_attention(si_ctx* ctx, const float* qkv, float* out, int B, int T, int H, int heads, hipStream_t st,
                        unsigned short* out16, bool att_bf16, const int32_t* valid_frames, const int32_t* row_off, double real_t2) {
    if (heads <= 0 || H != heads * 64) return si_fail(ctx, SI_EINVAL, "attention kernel needs head_dim 64 (H=%d heads=%d)", H, heads);
    if (B <= 0 || T <= 0) return SI_OK;
    dim3 grid((T + 127) / 128, B * heads);
    // bf16 encoder mode (out16 given): the bf16-MFMA form; SI_ATT_BF16=0 at context creation keeps the exact-fp32 kernel
    si_prof_begin(ctx, (out16 && att_bf16) ? "attention_bf16" : "attention_f32", 4.0 * (real_t2 > 0 ? real_t2 : B * (double)T * T) * H, 16.0 * B * T * H, st);   // 2*T^2*H MACs
    if (out16 && att_bf16) hipLaunchKernelGGL(attention_bf16_kernel, grid, dim3(256), 0, st, qkv, out16, T, H, heads, valid_frames, row_off);
    else hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, st, qkv, out, out16, T, H, heads, valid_frames, row_off);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ codebook
// arg-max with torch.argmax's rules: a NaN is the maximum, ties (and several NaNs) go to the lowest index.  The result is
// always a valid index: every thread seeds its running best with its first candidate unconditionally, threads without
// a candidate carry (-inf, INT_MAX) and lose every comparison against a real index.
__device__ __forceinline__ bool argmax_better(float a, int ai, float b, int bi) {
    const bool an = a != a, bn = b != b;
    if (an != bn) return an;
    if (!an && a != b) return a > b;
    return ai < bi;
}
// cosine arg-max of LDS-resident row v (D floats) against the centred codebook; every thread returns the label
__device__ __forceinline__ int codebook_argmax_128(const float* v, int D, const float* __restrict__ cc,
                                                   const float* __restrict__ rnorm, int K, float* bs, int* bi) {
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int k = threadIdx.x; k < K; k += 128) {
        const float* c = cc + (long)k * D;
        float dot = 0.f;
        for (int d = 0; d < D; ++d) dot = fmaf(v[d], c[d], dot);
        const float sim = dot * rnorm[k];                            // ||v|| is common to all k
        if (besti == 0x7fffffff || argmax_better(sim, k, best, besti)) { best = sim; besti = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (argmax_better(ob, oi, best, besti)) { best = ob; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bs[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    const int lab = argmax_better(bs[1], bi[1], bs[0], bi[0]) ? bi[1] : bi[0];
    return min(lab, K - 1);                                          // K >= 1: thread 0 always holds a real index
}

// One workgroup per (clip, masked frame).  D <= 128.
// frame_cnt (B) or null: ragged batches -- clip b replaces frame_cnt[b] frames (blind mode: all of ITS frames); j past it is label -1
__global__ __launch_bounds__(128) void codebook_splice_kernel(const float* __restrict__ feats, int T, int D,
                                                              const int32_t* __restrict__ frame_pos, int Lm,
                                                              const float* __restrict__ cc, const float* __restrict__ raw,
                                                              const float* __restrict__ rnorm, int K, float* __restrict__ mel,
                                                              int Tm, int64_t* __restrict__ labels, const int32_t* __restrict__ frame_cnt) {
    __shared__ float v[128];
    __shared__ float bs[2];
    __shared__ int bi[2];
    const int b = blockIdx.y, j = blockIdx.x;
    const int pos = frame_pos[b] + j;
    if (pos < 0 || pos >= T || (frame_cnt && j >= frame_cnt[b])) {   // uniform per block: no such encoder frame
        if (threadIdx.x == 0 && labels) labels[(long)b * Lm + j] = -1;
        return;
    }
    const float* f = feats + ((long)b * T + pos) * D;
    if ((int)threadIdx.x < D) v[threadIdx.x] = f[threadIdx.x];
    __syncthreads();
    const int lab = codebook_argmax_128(v, D, cc, rnorm, K, bs, bi);
    if (threadIdx.x == 0 && labels) labels[(long)b * Lm + j] = lab;
    if (pos < Tm && (int)threadIdx.x < D) mel[((long)b * D + threadIdx.x) * Tm + pos] = raw[(long)lab * D + threadIdx.x];
}

// Loss half of LossFunction.cos_sim + cos_sim_target_labels (I_ea/loss_fn.py:29-62).
// One workgroup per (clip, masked frame): term = 1 - cos(v, c_target), pred = arg-max_k cos(v, c_k) (same rule as the
// splice kernel), cpt = cos(c_pred, c_target); all cosines as F.cosine_similarity computes them: both vectors divided
// by max(norm, 1e-8) first, then the dot product.  A target label outside [0, K) yields NaN in both outputs.
__device__ __forceinline__ float block_sum_128(float x, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    __syncthreads();                                                 // scratch reuse across calls
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    return scratch[0] + scratch[1];
}

__global__ __launch_bounds__(128) void codebook_metrics_kernel(const float* __restrict__ feats, int T, int D,
                                                               const int32_t* __restrict__ frame_pos, int Lm,
                                                               const float* __restrict__ cc, const float* __restrict__ rnorm,
                                                               int K, const int64_t* __restrict__ target,
                                                               float* __restrict__ terms, int64_t* __restrict__ pred,
                                                               float* __restrict__ cos_pt) {
    __shared__ float v[128];
    __shared__ float red[2];
    __shared__ float bs[2];
    __shared__ int bi[2];
    const int b = blockIdx.y, j = blockIdx.x;
    const int pos = frame_pos[b] + j;
    const long o = (long)b * Lm + j;
    const long y = target[o];
    const float nanv = __builtin_nanf("");
    if (pos < 0 || pos >= T || y < 0 || y >= K) {                    // uniform per block
        if (threadIdx.x == 0) { terms[o] = nanv; cos_pt[o] = nanv; if (pred) pred[o] = -1; }
        return;
    }
    const int d = threadIdx.x;
    const float* f = feats + ((long)b * T + pos) * D;
    const float vd = d < D ? f[d] : 0.f;
    v[d] = vd;
    const float* ct = cc + y * D;
    const float cd = d < D ? ct[d] : 0.f;
    const float vn = fmaxf(sqrtf(block_sum_128(vd * vd, red)), 1e-8f);
    const float cn = fmaxf(sqrtf(block_sum_128(cd * cd, red)), 1e-8f);
    const float cosvt = block_sum_128((vd / vn) * (cd / cn), red);
    // arg-max over the centred codebook (the splice kernel's rule)
    const int lab = codebook_argmax_128(v, D, cc, rnorm, K, bs, bi);
    const float pd = d < D ? cc[(long)lab * D + d] : 0.f;
    const float pn = fmaxf(sqrtf(block_sum_128(pd * pd, red)), 1e-8f);
    const float cospt = block_sum_128((pd / pn) * (cd / cn), red);
    if (threadIdx.x == 0) {
        terms[o] = 1.0f - cosvt;                                     // -(cos - 1), loss_fn.py:41-43
        cos_pt[o] = cospt;
        if (pred) pred[o] = lab;
    }
}

// loss = sum of the per-frame terms, in a fixed order (deterministic): double partials, tree over 256 threads
__global__ __launch_bounds__(256) void sum_terms_kernel(const float* __restrict__ terms, long n, float* __restrict__ out) {
    __shared__ double part[256];
    double a = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) a += (double)terms[i];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)part[0];
}

int si_launch_codebook_metrics(si_ctx* ctx, const float* feats, int B, int T, int D, const int32_t* frame_pos, int Lm,
                               const float* cb_centered, const float* cb_rnorm, int K, const int64_t* target, float* terms,
                               float* loss, int64_t* pred, float* cos_pt, hipStream_t st) {
    if (D > 128) return si_fail(ctx, SI_EINVAL, "codebook dim %d > 128", D);
    if (B <= 0 || Lm <= 0) return SI_OK;
    si_prof_begin(ctx, "codebook_metrics", 2.0 * B * Lm * (double)K * D, 4.0 * B * Lm * 2.0 * D, st);
    hipLaunchKernelGGL(codebook_metrics_kernel, dim3(Lm, B), dim3(128), 0, st, feats, T, D, frame_pos, Lm, cb_centered, cb_rnorm, K,
                       target, terms, pred, cos_pt);
    hipLaunchKernelGGL(sum_terms_kernel, dim3(1), dim3(256), 0, st, terms, (long)B * Lm, loss);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// k-means unit assignment: label = argmin_k ||x - c_k||^2 = argmin_k (||c_k||^2 - 2 x.c_k), the
// quantity sklearn's KMeans.predict minimises (I_da/scripts/inpainting.py:204-205 calls it on HuBERT features; same
// distance as I_ea/dataset/km_label.py:20-24).  One workgroup per row; thread k walks centroid k (L2-resident table).
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const float* __restrict__ x, int D, const float* __restrict__ cent,
                                                            int K, int64_t* __restrict__ labels, float* __restrict__ dist) {
    extern __shared__ float xs[];
    __shared__ float bs[4];
    __shared__ int bi[4];
    const long row = blockIdx.x;
    const float* xr = x + row * D;
    float xx = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = xr[d]; xs[d] = v; xx = fmaf(v, v, xx); }
    __syncthreads();
    float best = INFINITY;
    int besti = 0x7fffffff;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float* c = cent + (long)k * D;
        float dot = 0.f, cc = 0.f;
        for (int d = 0; d < D; ++d) { const float cv = c[d]; dot = fmaf(xs[d], cv, dot); cc = fmaf(cv, cv, cc); }
        const float s = cc - 2.f * dot;
        if (s < best) { best = s; besti = k; }                      // ascending k per thread: first minimum wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        xx += __shfl_xor(xx, o, 64);
    }
    __shared__ float xparts[4];
    if ((threadIdx.x & 63) == 0) { bs[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; xparts[threadIdx.x >> 6] = xx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = bs[0]; int i = bi[0];
        for (int w = 1; w < 4; ++w) if (bs[w] < b || (bs[w] == b && bi[w] < i)) { b = bs[w]; i = bi[w]; }
        labels[row] = i;
        if (dist) dist[row] = b + (xparts[0] + xparts[1] + xparts[2] + xparts[3]);   // squared distance to the winner
    }
}

// The same assignment as a GEMM on the matrix pipe (exact-fp32 products: v_mfma_f32_32x32x2_f32) with the arg-min in its
// epilogue: rows x K x D = 6368 x 100 x 1024 per I_da call is 1.3 GFLOP, which the one-workgroup-per-row kernel above ran at
// 8 TFLOP/s (164 us: every row re-reads the whole centroid table through scalar-ish fp32 FMAs).  One workgroup = 32 rows x a block
// of up to 128 centroids; its four waves split the D reduction four ways (so that all four SIMDs of the CU work on the 199 row tiles of
// a call) and meet in LDS; then s_k = |c_k|^2 - 2 x.c_k and the first minimum, as above.  The dot products are exact-fp32
// products summed in another order than the scalar kernel's, so a label can differ from it only at a near-tie of the distances
// (the tests hold both kernels to the same near-tie rule against the oracle).  cnorm (K) = |c_k|^2 comes from a small kernel.
__global__ __launch_bounds__(256) void kmeans_cnorm_kernel(const float* __restrict__ cent, int D, float* __restrict__ cnorm) {
    const int k = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = cent[(long)k * D + d]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) cnorm[k] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void kmeans_mfma_kernel(const float* __restrict__ x, long rows, int D, const float* __restrict__ cent, int K,
                                                          const float* __restrict__ cnorm, int64_t* __restrict__ labels, float* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) char km_smem[];     // 66 KB: above the 64 KB a kernel gets without asking
    float (*dots)[32][132] = reinterpret_cast<float (*)[32][132]>(km_smem);   // per wave: partial x.c of 32 rows x 128 centroids (+4: bank spread)
    float (*xxs)[32] = reinterpret_cast<float (*)[32]>(km_smem + 4 * 32 * 132 * sizeof(float));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long row0 = (long)blockIdx.x * 32;
    const int Dw = D / 4;                                               // this wave's share of the reduction
    const long xrow = min(row0 + l31, rows - 1);                        // clamped: rows past the end are computed and dropped
    const float* xp = x + xrow * D + wave * Dw + half * (Dw / 2);
    float best = INFINITY;
    int besti = 0x7fffffff;
    float xx_row = 0.f;
    for (int kb = 0; kb < K; kb += 128) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float* cp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cp[j] = cent + (long)min(kb + 32 * j + l31, K - 1) * D + wave * Dw + half * (Dw / 2);
        float xx = 0.f;
        for (int s4 = 0; s4 < Dw / 8; ++s4) {                           // lane-half h walks floats [h Dw/2, (h+1) Dw/2) of the wave's range
            const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 4 * s4);
            f32x4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(cp[j] + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xx = fmaf(a[e], a[e], xx);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[j][e], acc[j], 0, 0, 0);
            }
        }
        // D[row][col]: this MFMA's A operand is the x row (M = rows), B the centroid (N = centroids): lane holds col = l31 of tile j,
        // rows (r & 3) + 8 (r >> 2) + 4 half
        __syncthreads();                                                // (a previous block's epilogue is done with `dots`)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dots[wave][(r & 3) + 8 * (r >> 2) + 4 * half][32 * j + l31] = acc[j][r];
        xx += __shfl_xor(xx, 32, 64);
        if (kb == 0 && half == 0) xxs[wave][l31] = xx;
        __syncthreads();
        // arg-min: 8 threads per row, each walks 16 of the block's 128 centroids in ascending order
        const int r = tid >> 3, c0 = tid & 7;
        if (kb == 0) xx_row = (xxs[0][r] + xxs[1][r]) + (xxs[2][r] + xxs[3][r]);
        for (int c = c0; c < 128; c += 8) {
            const int k = kb + c;
            if (k < K) {
                const float dot = (dots[0][r][c] + dots[1][r][c]) + (dots[2][r][c] + dots[3][r][c]);
                const float sv = cnorm[k] - 2.f * dot;
                if (sv < best) { best = sv; besti = k; }                // ascending k per thread: first minimum wins
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {                                   // the 8 threads of a row are 8 consecutive lanes
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const long row = row0 + (tid >> 3);
    if ((tid & 7) == 0 && row < rows) {
        labels[row] = besti;
        if (dist) dist[row] = best + xx_row;
    }
}

int si_launch_kmeans_assign(si_ctx* ctx, const float* x, long rows, int D, const float* cent, int K, int64_t* labels, float* dist,
                            hipStream_t st, float* cnorm_scratch) {
    if (D <= 0 || D > 8192 || K <= 0) return si_fail(ctx, SI_EINVAL, "kmeans_assign: D=%d (<= 8192) K=%d", D, K);
    if (rows <= 0) return SI_OK;
    if (cnorm_scratch && D % 32 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(cent) & 15) == 0) {
        const size_t km_lds = (4 * 32 * 132 + 4 * 32) * sizeof(float);
        if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kmeans_mfma_kernel), km_lds)) return rc;
        si_prof_begin(ctx, "kmeans_assign", 2.0 * rows * (double)K * D, 4.0 * rows * D, st);
        hipLaunchKernelGGL(kmeans_cnorm_kernel, dim3(K), dim3(256), 0, st, cent, D, cnorm_scratch);
        hipLaunchKernelGGL(kmeans_mfma_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), km_lds, st, x, rows, D, cent, K, cnorm_scratch, labels, dist);
        si_prof_end(ctx, st);
        SI_HIP_CHECK(hipGetLastError());
        return SI_OK;
    }
    si_prof_begin(ctx, "kmeans_assign", 2.0 * rows * (double)K * D, 4.0 * rows * D, st);
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3((unsigned)rows), dim3(256), (size_t)D * sizeof(float), st, x, D, cent, K, labels, dist);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// I_da code splice (I_da/scripts/inpainting.py:209-214): the masked stream's units survive only inside the mask,
// `code_inpainting[: fs // hop] = code[: fs // hop]; code_inpainting[(fs + ms) // hop :] = code[(fs + ms) // hop :]`.
__global__ __launch_bounds__(256) void code_splice_kernel(const int64_t* __restrict__ clean, const int64_t* __restrict__ masked,
                                                          const int32_t* __restrict__ first, const int32_t* __restrict__ last, int T,
                                                          int64_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const long i = (long)b * T + t;
    out[i] = (t < first[b] || t >= last[b]) ? clean[i] : masked[i];
}

int si_launch_code_splice(si_ctx* ctx, const int64_t* clean, const int64_t* masked, const int32_t* first, const int32_t* last, int B, int T,
                          int64_t* out, hipStream_t st) {
    if (B <= 0 || T <= 0) return SI_OK;
    si_prof_begin(ctx, "code_splice", 0.0, 24.0 * B * T, st);
    hipLaunchKernelGGL(code_splice_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, clean, masked, first, last, T, out);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// expected_inpaint splice: given labels -> raw centroids into the mel (I_ea/predict.py:177-189)
__global__ __launch_bounds__(128) void codebook_gather_kernel(const int64_t* __restrict__ labels, int D,
                                                              const int32_t* __restrict__ frame_pos, int Lm,
                                                              const float* __restrict__ raw, int K, float* __restrict__ mel, int Tm) {
    const int b = blockIdx.y, j = blockIdx.x;
    const int pos = frame_pos[b] + j;
    const long lab = labels[(long)b * Lm + j];
    if (pos < 0 || pos >= Tm || lab < 0 || lab >= K) return;
    if ((int)threadIdx.x < D) mel[((long)b * D + threadIdx.x) * Tm + pos] = raw[lab * D + threadIdx.x];
}

int si_launch_codebook_gather(si_ctx* ctx, const int64_t* labels, int B, int D, const int32_t* frame_pos, int Lm,
                              const float* cb_raw, int K, float* mel, int Tm, hipStream_t st) {
    if (D > 128) return si_fail(ctx, SI_EINVAL, "codebook dim %d > 128", D);
    if (B <= 0 || Lm <= 0) return SI_OK;
    si_prof_begin(ctx, "codebook_gather", 0.0, 8.0 * B * Lm * D, st);
    hipLaunchKernelGGL(codebook_gather_kernel, dim3(Lm, B), dim3(128), 0, st, labels, D, frame_pos, Lm, cb_raw, K, mel, Tm);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_codebook_splice(si_ctx* ctx, const float* feats, int B, int T, int D, const int32_t* frame_pos, int Lm,
                              const float* cb_centered, const float* cb_raw, const float* cb_rnorm, int K, float* mel, int Tm,
                              int64_t* labels, hipStream_t st, const int32_t* frame_cnt) {
    if (D > 128) return si_fail(ctx, SI_EINVAL, "codebook dim %d > 128", D);
    if (B <= 0 || Lm <= 0) return SI_OK;
    si_prof_begin(ctx, "codebook_splice", 2.0 * B * Lm * (double)K * D, 4.0 * B * Lm * 2.0 * D, st);
    hipLaunchKernelGGL(codebook_splice_kernel, dim3(Lm, B), dim3(128), 0, st, feats, T, D, frame_pos, Lm, cb_centered, cb_raw,
                       cb_rnorm, K, mel, Tm, labels, frame_cnt);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
