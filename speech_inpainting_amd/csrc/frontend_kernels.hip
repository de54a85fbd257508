// frontend_kernels.hip -- log-mel front-end of the vocoder side, gfx950.
//
// Replaces, on the predict path, I_ea/predict.py:99-106 (zero the masked span of the 22.05 kHz clip,
// `librosa.util.normalize(x) * 0.95`) and I_ea/dataset/mel_dump.py:40-98 (`get_mel`: reflect-pad 312, STFT with
// n_fft = win = 1024, hop 441, periodic Hann, center=False; sqrt(re^2 + im^2 + 1e-9); 80-band Slaney mel basis;
// log(clamp(., 1e-5))).
//
// Shape of the work: per clip 200 frames x 1024 samples x 1026 DFT outputs = 0.2 GMAC, 0.2 % of the path, so the DFT
// is simply one exact-fp32 tap-GEMM launch over an explicit frame matrix (the frames of a 32-clip batch are 26 MB:
// writing them once costs microseconds and keeps every GEMM row 16-byte aligned; hop 441 is odd).  The three kernels
// here are the HBM/latency-side pieces around that GEMM:
//   wave_peak_kernel    max |x| of the masked clip                                   (one pass over 353 KB / clip)
//   mel_frames_kernel   mask -> normalise -> reflect-pad -> window, written as the (B*Tm, 1024) frame matrix
//   mel_project_kernel  magnitude, banded mel projection (only the non-zero span of each triangle), log, transposed
//                       store into the reference's (B, 80, Tm) layout
#include "common.h"

// n_len (B) or null: ragged batches -- clip b holds n_len[b] samples inside its row of Ns
__global__ __launch_bounds__(1024) void wave_peak_kernel(const float* __restrict__ wav, const int32_t* __restrict__ ms,
                                                         const int32_t* __restrict__ me, int Ns, float* __restrict__ peak,
                                                         const int32_t* __restrict__ n_len) {
    const int b = blockIdx.x;
    const float* x = wav + (size_t)b * Ns;
    const int N = n_len ? n_len[b] : Ns;
    const int s = ms ? ms[b] : 0, e = ms ? me[b] : 0;
    float m = 0.f;
    // 16 bytes per lane, four loads in flight (one workgroup walks a whole clip); scalar loop for the tail / unaligned clips
    const bool vec = (Ns & 3) == 0 && (reinterpret_cast<size_t>(wav) & 15) == 0;
    const int n4 = vec ? (N & ~3) : 0;
#pragma unroll 4
    for (int i = threadIdx.x * 4; i < n4; i += 1024 * 4) {
        const float4 q = *reinterpret_cast<const float4*>(x + i);
        const float qe[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) m = fmaxf(m, (i + t >= s && i + t < e) ? 0.f : fabsf(qe[t]));
    }
    for (int i = n4 + threadIdx.x; i < N; i += 1024) {
        const float v = (i >= s && i < e) ? 0.f : fabsf(x[i]);
        m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, part[w]);
        peak[b] = m;
    }
}

// grid (Tm, B), 256 threads.  The windowed frame w[k] = x[.] * hann[k] is written FOLDED for the two half-size DFT GEMMs (api.hip):
// [ s_0 .. s_{n/2} | zeros up to kc | 0, d_1 .. d_{n/2-1} ],  s_k = w[k] + w[n - k], d_k = w[k] - w[n - k]  (s_0 = w[0], s_{n/2} = w[n/2]).
// n_len / tm_len (B) or null: ragged batches -- clip b holds n_len[b] samples (row stride Ns; the reflection is at ITS end) and
// tm_len[b] frames (frame-matrix stride Tm); frames past them are not written
__global__ __launch_bounds__(256) void mel_frames_kernel(const float* __restrict__ wav, const int32_t* __restrict__ ms,
                                                         const int32_t* __restrict__ me, const float* __restrict__ peak,
                                                         const float* __restrict__ hann, int Ns, int Tm, int hop, int pad,
                                                         int nfft, int kc, int normalize, float* __restrict__ frames,
                                                         const int32_t* __restrict__ n_len, const int32_t* __restrict__ tm_len) {
    const int m = blockIdx.x, b = blockIdx.y;
    if (tm_len && m >= tm_len[b]) return;
    const float* x = wav + (size_t)b * Ns;
    const int N = n_len ? n_len[b] : Ns;
    const int s = ms ? ms[b] : 0, e = ms ? me[b] : 0;
    // librosa.util.normalize: divide by max |x|; a peak below the smallest normal float leaves the clip unscaled
    const float pk = normalize ? peak[b] : 1.f;
    const float div = pk < 1.17549435e-38f ? 1.f : pk;
    const int half = nfft / 2, flen = kc + half;
    float* dst = frames + ((size_t)b * Tm + m) * flen;
    auto win = [&](int k) {                                     // windowed sample k of this frame
        int j = m * hop + k - pad;                              // position in the un-padded clip
        if (j < 0) j = -j;                                      // reflect (no edge repeat), mel_dump.py:72
        if (j >= N) j = 2 * (N - 1) - j;
        float v = (j >= s && j < e) ? 0.f : x[j];
        if (normalize) v = (v / div) * 0.95f;                   // predict.py:104, in the script's operation order
        return v * hann[k];
    };
    for (int q = threadIdx.x * 4; q < flen; q += 1024) {
        float4 o;
        float* op = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = q + u;
            float v = 0.f;
            if (i <= half) v = (i == 0 || i == half) ? win(i) : win(i) + win(nfft - i);
            else if (i > kc) v = win(i - kc) - win(nfft - (i - kc));            // (i == kc: d_0 = 0; half < i < kc: padding)
            op[u] = v;
        }
        *reinterpret_cast<float4*>(dst + q) = o;
    }
}

// One workgroup per frame.  spec row = [re(0..nbin-1) | pad | im(0..nbin-1) at im_off] as the two DFT GEMMs wrote it.
// basis_t is the mel basis transposed to (nbin, nmel); band i is non-zero on bins [lo[i], hi[i]).
__global__ __launch_bounds__(128) void mel_project_kernel(const float* __restrict__ spec, int ld_spec, int nbin,
                                                          const float* __restrict__ basis_t, const int32_t* __restrict__ lo,
                                                          const int32_t* __restrict__ hi, int nmel, int Tm, int im_off,
                                                          float* __restrict__ mel, const int32_t* __restrict__ tm_len) {
    extern __shared__ float mag[];
    const long row = blockIdx.x;
    if (tm_len && (int)(row % Tm) >= tm_len[row / Tm]) {               // ragged batches: a frame past the clip's own is written as zero
        for (int i = threadIdx.x; i < nmel; i += blockDim.x) mel[((size_t)(row / Tm) * nmel + i) * Tm + row % Tm] = 0.f;
        return;
    }
    const float* sp = spec + row * ld_spec;
    for (int f = threadIdx.x; f < nbin; f += blockDim.x) {
        const float re = sp[f], im = sp[im_off + f];
        mag[f] = sqrtf(re * re + im * im + 1e-9f);              // mel_dump.py:89
    }
    __syncthreads();
    const int b = (int)(row / Tm), m = (int)(row % Tm);
    for (int i = threadIdx.x; i < nmel; i += blockDim.x) {
        float acc = 0.f;
        for (int f = lo[i]; f < hi[i]; ++f) acc = fmaf(basis_t[(size_t)f * nmel + i], mag[f], acc);
        mel[((size_t)b * nmel + i) * Tm + m] = logf(fmaxf(acc, 1e-5f));   // mel_dump.py:31,91
    }
}

int si_launch_wave_peak(si_ctx* ctx, const float* wav, const int32_t* ms, const int32_t* me, int B, int N, float* peak,
                        hipStream_t st, const int32_t* n_len) {
    si_prof_begin(ctx, "wave_peak", (double)B * N, (double)B * N * 4, st);
    wave_peak_kernel<<<B, 1024, 0, st>>>(wav, ms, me, N, peak, n_len);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_mel_frames(si_ctx* ctx, const float* wav, const int32_t* ms, const int32_t* me, const float* peak,
                         const float* hann, int B, int N, int Tm, int hop, int pad, int nfft, int kc, int normalize, float* frames,
                         hipStream_t st, const int32_t* n_len, const int32_t* tm_len) {
    if (nfft % 8 || kc % 4 || kc <= nfft / 2) return si_fail(ctx, SI_EINVAL, "mel_frames: n_fft %d / folded width %d unsupported", nfft, kc);
    if (N <= pad) return si_fail(ctx, SI_EINVAL, "mel_frames: clip of %d samples is not longer than the reflect pad %d", N, pad);
    si_prof_begin(ctx, "mel_frames", 3.0 * B * Tm * nfft, (double)B * N * 4 + (double)B * Tm * nfft * 4, st);
    mel_frames_kernel<<<dim3(Tm, B), 256, 0, st>>>(wav, ms, me, peak, hann, N, Tm, hop, pad, nfft, kc, normalize, frames, n_len, tm_len);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

int si_launch_mel_project(si_ctx* ctx, const float* spec, int ld_spec, int nbin, int im_off, const float* basis_t, const int32_t* lo,
                          const int32_t* hi, int nmel, int B, int Tm, float* mel, hipStream_t st, const int32_t* tm_len) {
    const long rows = (long)B * Tm;
    si_prof_begin(ctx, "mel_project", (double)rows * (4.0 * nbin + 2.0 * nbin * 2), (double)rows * (2.0 * nbin + nmel) * 4, st);
    mel_project_kernel<<<(unsigned)rows, 128, (size_t)nbin * sizeof(float), st>>>(spec, ld_spec, nbin, basis_t, lo, hi, nmel, Tm, im_off, mel, tm_len);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ f-3: resampler
// Polyphase FIR resampling y = (x up-sampled by `up`) * h, down-sampled by `down` -- scipy.signal.upfirdn's definition,
// with resample_poly's centring (the taps arrive already zero-pre-padded; `pre_remove` leading outputs are dropped):
//     y[m'] = sum_j h[(m' + pre_remove) * down - j * up] * x[j]        over the j that keep the tap index in [0, ntaps)
// One thread per output sample; the tap table sits in LDS (8.8 k taps for 22.05 kHz -> 16 kHz), a thread walks every
// `up`-th tap from its phase while its input index walks down -- neighbouring threads read neighbouring inputs.
__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, int n_in, const float* __restrict__ taps,
                                                            int ntaps, int up, int down, int pre_remove, int n_out,
                                                            float* __restrict__ y) {
    extern __shared__ float hs[];
    for (int i = threadIdx.x; i < ntaps; i += 256) hs[i] = taps[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_out) return;
    const long t = (long)(m + pre_remove) * down;
    long j = t / up;                                  // newest input that can touch this output
    int k = (int)(t - j * up);                        // its tap
    if (j > n_in - 1) { const long skip = j - (n_in - 1); k += (int)(skip * up); j = n_in - 1; }
    const float* xb = x + (long)b * n_in;
    float acc = 0.f;
    for (; k < ntaps && j >= 0; k += up, --j) acc = fmaf(hs[k], xb[j], acc);
    y[(long)b * n_out + m] = acc;
}

int si_launch_resample_poly(si_ctx* ctx, const float* x, int B, int n_in, const float* taps, int ntaps, int up, int down,
                            int pre_remove, int n_out, float* y, hipStream_t st) {
    if (up < 1 || down < 1 || ntaps < 1 || (size_t)ntaps * 4 > 150 * 1024)
        return si_fail(ctx, SI_EINVAL, "resample: up=%d down=%d ntaps=%d (tap table must fit 150 KB of LDS)", up, down, ntaps);
    if (B <= 0 || n_out <= 0) return SI_OK;
    auto kern = resample_poly_kernel;
    const size_t lds = (size_t)ntaps * sizeof(float);
    if (int rc = si_ensure_dyn_lds(ctx, reinterpret_cast<const void*>(kern), lds)) return rc;
    si_prof_begin(ctx, "resample_poly", 2.0 * B * n_out * ((double)ntaps / up), 4.0 * B * ((double)n_in + n_out), st);
    hipLaunchKernelGGL(kern, dim3((n_out + 255) / 256, B), dim3(256), lds, st, x, n_in, taps, ntaps, up, down, pre_remove, n_out, y);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ f-3: librosa's resampler
// `librosa.load(path, sr=16000)` of librosa 0.9.1 (I_ea/predict.py:79-80; requirements.txt:3) resamples with resampy's `kaiser_best`
// filter: band-limited interpolation with a table of the windowed sinc's right wing (512 samples per zero crossing, 64 zero
// crossings) and linear interpolation between table entries.  Per output sample t (resampy/interpn.py):
//     time = time_register[t] (1 / ratio accumulated by repeated addition in float64 -- the HOST builds that very sequence, because
//            where the exact time is an integer its rounding picks between two table phases that give different samples);
//     n = int(time);  frac = scale * (time - n);  off, eta = split(frac * num_table)
//     y[t]  = sum_{i < min(n + 1, (nwin - off) / step)}         (win[off + i step] + eta dwin[off + i step]) * x[n - i]
//           + sum_{k < min(n_in - n - 1, (nwin - off') / step)}  (win[off' + k step] + eta' dwin[...])       * x[n + 1 + k],   off', eta' from scale - frac
// for t < int(n_in * ratio); librosa then pads with zeros (util.fix_length).  Weights and sums in float64 (the tables are float64):
// the result reproduces the reference-held 16 kHz rendering of LJ001-0001 bit for bit after its int16 quantisation.
// One thread per output sample; the 0.5 MB of tables sit in L2 and the 64 lanes of a load share ~24 cache lines (their phases fall
// inside one `step` of the table).  (A phase-major copy of the table -- a thread's taps consecutive in memory -- was measured: 2.9 ms
// instead of 0.92 per 32 x 4 s: every lane then owns its own lines.)  n_len (B) or null: ragged batches, clip b holds n_len[b] samples.
__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, const int32_t* __restrict__ n_len, int n_in_stride,
                                                            const double* __restrict__ win, const double* __restrict__ dwin, int nwin,
                                                            int num_table, int step, double scale, double ratio,
                                                            const double* __restrict__ time_reg, int n_out_stride, float* __restrict__ y) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_out_stride) return;
    const int n_in = n_len ? n_len[b] : n_in_stride;
    const int n_out = (int)((double)n_in * ratio);
    float* yo = y + (long)b * n_out_stride + t;
    if (t >= n_out) { *yo = 0.f; return; }
    const float* xb = x + (long)b * n_in_stride;
    const double time = time_reg[t];
    const int n = (int)time;
    double frac = scale * (time - n);
    double acc = 0.0;
    {
        const double idx = frac * num_table;
        const int off = (int)idx;
        const double eta = idx - off;
        const int cnt = min(n + 1, (nwin - off) / step);
        for (int i = 0; i < cnt; ++i) {
            const int w = off + i * step;
            acc += (win[w] + eta * dwin[w]) * (double)xb[n - i];
        }
    }
    {
        frac = scale - frac;
        const double idx = frac * num_table;
        const int off = (int)idx;
        const double eta = idx - off;
        const int cnt = min(n_in - n - 1, (nwin - off) / step);
        for (int k = 0; k < cnt; ++k) {
            const int w = off + k * step;
            acc += (win[w] + eta * dwin[w]) * (double)xb[n + k + 1];
        }
    }
    *yo = (float)acc;
}

int si_launch_resample_sinc(si_ctx* ctx, const float* x, const int32_t* n_len, int B, int n_in, const double* win, const double* dwin, int nwin,
                            int num_table, int step, double scale, double ratio, const double* time_reg, int n_out, float* y, hipStream_t st) {
    if (nwin < 2 || num_table < 1 || step < 1 || !(scale > 0.0) || !(ratio > 0.0))
        return si_fail(ctx, SI_EINVAL, "resample_sinc: bad filter table (nwin %d, num_table %d, step %d)", nwin, num_table, step);
    if (B <= 0 || n_out <= 0) return SI_OK;
    si_prof_begin(ctx, "resample_sinc", 4.0 * B * n_out * (2.0 * nwin / step), 4.0 * B * ((double)n_in + n_out), st);
    hipLaunchKernelGGL(resample_sinc_kernel, dim3((n_out + 255) / 256, B), dim3(256), 0, st, x, n_len, n_in, win, dwin, nwin, num_table, step, scale, ratio,
                       time_reg, n_out, y);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}

// ------------------------------------------------------------------------------------------------ B6: int16 PCM
// The script's `audio * MAX_WAV_VALUE` + `.astype('int16')` (I_ea/predict.py:204-206): the fp32 product truncated toward zero.  The
// generator ends in tanh, so the product lies in [-32768, 32768]; 32768.0 (an exactly saturated sample) is outside int16, where the
// reference's cast is undefined behaviour: pinned to 32767 (audio.to_int16_pcm does the same on the host).  NaN -> 0.
__global__ __launch_bounds__(256) void pcm16_kernel(const float* __restrict__ wav, long n, int16_t* __restrict__ out) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n && (reinterpret_cast<size_t>(wav) & 15) == 0 && (reinterpret_cast<size_t>(out) & 7) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(wav + i);
        const float q[4] = {v.x, v.y, v.z, v.w};
        short r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = q[e] * 32768.0f;
            r[e] = (short)(int)fminf(fmaxf(truncf(p == p ? p : 0.f), -32768.f), 32767.f);
        }
        *reinterpret_cast<short4*>(out + i) = make_short4(r[0], r[1], r[2], r[3]);
        return;
    }
    for (long j = i; j < n && j < i + 4; ++j) {
        const float p = wav[j] * 32768.0f;
        out[j] = (int16_t)(int)fminf(fmaxf(truncf(p == p ? p : 0.f), -32768.f), 32767.f);
    }
}

int si_launch_pcm16(si_ctx* ctx, const float* wav, long n, int16_t* out, hipStream_t st) {
    if (n <= 0) return SI_OK;
    si_prof_begin(ctx, "pcm16", (double)n, 6.0 * n, st);
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, wav, n, out);
    si_prof_end(ctx, st);
    SI_HIP_CHECK(hipGetLastError());
    return SI_OK;
}
