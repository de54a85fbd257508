// metrics_kernels.hip -- evaluation metrics of the inpainting path on the GPU (the mel / waveform
// half; the codeword half is codebook_metrics_kernel in encoder_kernels.hip).  gfx950, wave64.
//
//  * mel_metrics : `Metrics.avg_cosine_sim`, `.avg_d2_dist`, `.rmse` (I_ea/metrics.py:38-62) on two mel segments
//    (D bins, L frames) per clip: cosine over the bins of the centred frames, averaged over frames; the bin-mean-removed
//    log-spectral distance per frame (20 / ln 10 * sqrt(mean_d diff^2)) averaged over frames; and its global form.
//  * sisdr       : `Metrics.sisdr` (I_ea/metrics.py:127-142): scale-invariant SDR of an estimate against a reference.
// One workgroup per clip; sums in fp64 with a fixed reduction order (deterministic).
#include "common.h"

__device__ __forceinline__ double mk_block_sum(double v, double* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                                                 // scratch reuse across calls
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += scratch[i];
    return s;
}

// a, b: (B, D, L) channels-first (the layout of every mel in this library); out[b] = {avg_cosine_sim, avg_d2_dist, rmse}
__global__ __launch_bounds__(256) void mel_metrics_kernel(const float* __restrict__ a, const float* __restrict__ b, int D, int L,
                                                          const float* __restrict__ center, float* __restrict__ out) {
    __shared__ double scratch[4];
    const int clip = blockIdx.x;
    const float* pa = a + (long)clip * D * L;
    const float* pb = b + (long)clip * D * L;
    double cos_sum = 0.0, d2_sum = 0.0, sq_sum = 0.0;
    for (int l = threadIdx.x; l < L; l += 256) {
        float dot = 0.f, na = 0.f, nb = 0.f, ma = 0.f, mb = 0.f;
        for (int d = 0; d < D; ++d) {
            const float x = pa[(long)d * L + l], y = pb[(long)d * L + l];
            const float c = center ? center[d] : 0.f;
            dot = fmaf(x - c, y - c, dot); na = fmaf(x - c, x - c, na); nb = fmaf(y - c, y - c, nb);
            ma += x; mb += y;
        }
        // F.cosine_similarity(dim = 0, eps = 1e-8): x.y / (max(|x|, eps) * max(|y|, eps))
        cos_sum += (double)(dot / (fmaxf(sqrtf(na), 1e-8f) * fmaxf(sqrtf(nb), 1e-8f)));
        ma /= D; mb /= D;                                           // torch.mean(tensor, dim = 0): over the bins of this frame
        float sq = 0.f;
        for (int d = 0; d < D; ++d) {
            const float e = (pa[(long)d * L + l] - ma) - (pb[(long)d * L + l] - mb);
            sq = fmaf(e, e, sq);
        }
        d2_sum += (double)sqrtf(sq / D);
        sq_sum += (double)sq;
    }
    cos_sum = mk_block_sum(cos_sum, scratch);
    d2_sum = mk_block_sum(d2_sum, scratch);
    sq_sum = mk_block_sum(sq_sum, scratch);
    if (threadIdx.x == 0) {
        const double log_scale = 20.0 / log(10.0);
        out[3 * clip + 0] = (float)(cos_sum / L);
        out[3 * clip + 1] = (float)(log_scale * d2_sum / L);
        out[3 * clip + 2] = (float)(log_scale * sqrt(sq_sum / ((double)D * L)));
    }
}

// est, ref: (B, n); out[b] = 10 log10((eps + |a ref|^2) / (eps + |est - a ref|^2)), a = (eps + ref.est) / (|ref|^2 + eps),
// eps = the float32 machine epsilon (np.finfo(x_est.dtype).eps for float32 waveforms)
__global__ __launch_bounds__(1024) void sisdr_kernel(const float* __restrict__ est, const float* __restrict__ ref, int n, float* __restrict__ out) {
    __shared__ double scratch[16];
    const int clip = blockIdx.x;
    const float* e = est + (long)clip * n;
    const float* r = ref + (long)clip * n;
    double rr = 0.0, re = 0.0, ee = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double x = e[i], y = r[i];
        rr += y * y; re += y * x; ee += x * x;
    }
    rr = mk_block_sum(rr, scratch);
    re = mk_block_sum(re, scratch);
    ee = mk_block_sum(ee, scratch);
    if (threadIdx.x == 0) {
        const double eps = 1.1920928955078125e-07;
        const double a = (eps + re) / (rr + eps);
        const double sss = a * a * rr;
        const double snn = ee - 2.0 * a * re + a * a * rr;           // |est - a ref|^2
        out[clip] = (float)(10.0 * log10((eps + sss) / (eps + (snn > 0.0 ? snn : 0.0))));
    }
}

int si_launch_mel_metrics(si_ctx* ctx, const float* a, const float* b, int B, int D, int L, const float* center, float* out, hipStream_t st) {
    if (B <= 0 || D <= 0 || L <= 0) return si_fail(ctx, SI_EINVAL, "mel_metrics: empty input");
    si_prof_begin(ctx, "mel_metrics", 10.0 * B * D * (double)L, 16.0 * B * D * (double)L, st);
    hipLaunchKernelGGL(mel_metrics_kernel, dim3(B), dim3(256), 0, st, a, b, D, L, center, out);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_sisdr(si_ctx* ctx, const float* est, const float* ref, int B, int n, float* out, hipStream_t st) {
    if (B <= 0 || n <= 0) return si_fail(ctx, SI_EINVAL, "sisdr: empty input");
    si_prof_begin(ctx, "sisdr", 6.0 * B * (double)n, 8.0 * B * (double)n, st);
    hipLaunchKernelGGL(sisdr_kernel, dim3(B), dim3(1024), 0, st, est, ref, n, out);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
