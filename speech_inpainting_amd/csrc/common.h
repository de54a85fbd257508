// Internal declarations shared by the HIP translation units of libsi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/si_hip.h"

#define SI_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return si_fail_hip(ctx, _e, #expr, __FILE__, __LINE__);           \
    } while (0)

int si_fail_hip(si_ctx* ctx, hipError_t e, const char* what, const char* file, int line);
int si_fail(si_ctx* ctx, int code, const char* fmt, ...);
// HIP-event bracket around one kernel launch (no-op unless si_profile_start armed the context)
void si_prof_begin(si_ctx* ctx, const char* name, double flops, double bytes, hipStream_t st);
void si_prof_end(si_ctx* ctx, hipStream_t st);
// Diagnostic (SI_PROF_SHAPES=1 in the environment): the family name with the GEMM shape appended, so that a per-kernel table
// separates the launches of one instantiation; otherwise `name` itself.  The returned pointer is valid until the next call.
const char* si_prof_shape_name(const char* name, long M, int N, int K);
// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) when a launch needs more than the
// 64 KB default.  The high-water mark is kept per context (= per device): a process-wide cache would skip the call
// for a second context on another GPU.
int si_ensure_dyn_lds(si_ctx* ctx, const void* kern, size_t bytes);
int si_num_cus(si_ctx* ctx);   // compute units of the context's device (persistent-kernel grids)

typedef unsigned short bf16_t;   // raw bf16 bits

// ------------------------------------------------------------------------------------------------
// Ragged batches (BASELINE configs[4]: clips of different lengths in ONE launch, each clip's result equal to that clip run
// alone -- the reference's one-file-per-run script, I_ea/predict.py:76-207).  Storage keeps a fixed stride per clip (the
// longest clip's); a device int32 array holds every clip's own row count, and the persistent kernels number their tiles clip
// by clip WITHOUT gaps: tile t -> (clip, first row, the clip's rows).  `t` is wave-uniform; every lane gets the result.
// One vector load, a 6-step wave scan and a ballot per call (per tile of tens of microseconds).
// ------------------------------------------------------------------------------------------------
struct SiVlTile { int b, row0, L; };
#ifdef __HIPCC__
__device__ __forceinline__ SiVlTile si_vl_tile(const int32_t* __restrict__ lens, int B, int step, int t) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int base = 0;
    for (int s0 = 0; s0 < B; s0 += 64) {
        const int s = s0 + lane;
        const int Lb = s < B ? lens[s] : 0;
        const int nb = Lb > 0 ? (Lb + step - 1) / step : 0;
        int incl = nb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int tot = __shfl(incl, 63, 64);
        if (t < base + tot) {                                          // wave-uniform
            const unsigned long long m = __ballot(base + incl > t);
            const int l = (int)__builtin_ctzll(m);
            SiVlTile r;
            r.b = s0 + l;
            r.row0 = (t - base - __shfl(incl - nb, l, 64)) * step;
            r.L = __shfl(Lb, l, 64);
            return r;
        }
        base += tot;
    }
    return SiVlTile{B, 0, 0};                                          // t past the last tile
}
#endif
// the same count on the host: tiles of `step` rows over clips of lens[b] rows
static inline long si_vl_tiles(const int32_t* lens, int B, int step) {
    long n = 0;
    for (int b = 0; b < B; ++b) if (lens[b] > 0) n += (lens[b] + step - 1) / step;
    return n;
}

// ------------------------------------------------------------------------------------------------
// "Tap GEMM": every contraction of the path (Conv1d with stride/dilation/groups, ConvTranspose1d split
// into stride phases, Linear) is
//     out[seg][m][n] = epi( sum_tap sum_ci  pro(x[seg][m*stride + tap*dil - pad][g*Cin + ci]) * W[g][tap][n][ci] )
// on channels-last activations, i.e. a GEMM whose A operand is a shifted view of the activation matrix.
// ------------------------------------------------------------------------------------------------
enum { SI_ACT_NONE = 0, SI_ACT_GELU = 1 };

struct TapGemmParams {
    const float* x;        // [nseg][Lin][ldx] fp32, channels-last (NULL when x16 is given)
    const unsigned short* x16;  // same geometry, 16-bit in the math mode's operand type; pro_slope (if != 1) is applied to it while staging
    const void* w;         // [groups][ntaps][Npad][Cin]  (fp32, or bf16 hi plane)
    const void* w_lo;      // bf16x3: lo plane, same layout
    const float* bias;     // [groups*N] or nullptr
    const float* res;      // residual, indexed like out, or nullptr
    // res_stats != NULL: the residual is LayerNorm(res) -- res holds the rows BEFORE the normalisation, res_stats (mean, rstd) per row,
    // res_gamma / res_beta its affine pair (the normalised rows were written as a bf16 GEMM operand only); bf16 GEMM kernels only
    const float* res_stats; const float* res_gamma; const float* res_beta;
    const unsigned short* res16;  // residual as raw 16-bit values of the math mode's type (instead of res)
    int acc16;             // accumulate reads the previous value from out16 (raw 16-bit) instead of out
    float* out;            // fp32 output (may be NULL when only out16 is wanted)
    unsigned short* out16; // optional operand-ready copy for the consumer: type16(leaky_relu(v, out16_slope)), indexed like out
    float out16_slope;     // the consumer's prologue slope (1 = identity)
    int nseg, Lin, M;      // segments (clips), input rows and output rows per segment
    int ldx;               // input row stride (floats)
    long x_seg_stride;     // floats between segments
    int Cin;               // input channels per group (multiple of BK)
    int N, Npad;           // output columns per group, padded row count of W
    int ntaps, stride, dil, pad;
    int groups;
    int ldo;               // output row stride (floats)
    long o_seg_stride;
    long ooff;             // flat offset added to m*ldo + n (ConvTranspose phase layout: -pad*Cout)
    long olimit;           // stores outside [0, olimit) of a segment are dropped
    float pro_slope;       // leaky-relu slope applied to x while staging (1 = identity)
    int act;               // SI_ACT_*
    float alpha;           // v = (acc + bias -> act -> + res) * alpha
    int accumulate;        // out = v + out
    double algo_macs;      // algorithmic multiply-accumulates of the layer (0: derive from the GEMM shape)
    int wide_epilogue;     // set by the launcher: row-contiguous 16-byte epilogue through an LDS transpose
    int lingemm;           // caller allows the dedicated bf16 GEMM kernel (lingemm.hip) when the shape fits it
    // ragged batches (all device int32 (nseg), all NULL for uniform segments): segment s reads seg_lin[s] input rows (instead
    // of Lin: rows outside read as zero -- the convolution's padding at the clip's OWN end), computes seg_m[s] output rows
    // (instead of M: tiles past them exit at once) and stores inside [0, seg_orows[s] * olim_mul) (instead of olimit).
    // seg_row_off: PACKED rows -- segment s starts at row seg_row_off[s] of x (row stride ldx) and of out / res / out16
    // (row stride ldo) instead of at s * x_seg_stride / s * o_seg_stride.
    const int32_t* seg_lin;
    const int32_t* seg_m;
    const int32_t* seg_orows;
    int olim_mul;
    const int32_t* seg_row_off;
    const int32_t* seg_m_host;   // host copy of seg_m (grid sizing of the kernels that walk tiles without gaps)
};

// N-tile width the launcher uses for a given N; the packer pads W rows to a multiple of it.
static inline int si_pick_bn(int N) { return N >= 128 ? 128 : (N > 32 ? 64 : 32); }
static inline int si_round_up(int a, int b) { return (a + b - 1) / b * b; }

int si_launch_tapgemm(si_ctx* ctx, int math, const TapGemmParams& p, hipStream_t st);

// The encoder's bf16 GEMM (lingemm.hip): out[seg][m][n] = epi(sum_k A[seg][m][k] W[n][k] + bias[n]); A row m = K
// consecutive bf16 at x16 + seg * x_seg_stride + m * lda (lda < K: overlapping rows = a strided convolution on
// channels-last activations); W = ntaps blocks [N][Cin] (k = tap * Cin + ci).  Returns 1 when the shape is not covered.
struct LinGemmParams {
    const unsigned short* x16; int x_bytes;     // activations (bf16) and the size of their buffer in bytes
    int lda; long x_seg_stride;                 // elements between consecutive A rows / segments
    int nseg, M, K;
    const unsigned short* w; int w_bytes;
    int N, Cin, ntaps; long w_tap_stride;       // elements between tap blocks of W
    const float* bias; const float* res;        // res: fp32, indexed like out
    const float* res_stats; const float* res_gamma; const float* res_beta;   // see TapGemmParams
    float* out; unsigned short* out16;          // fp32 and / or bf16 output
    int ldo; long o_seg_stride;
    int act;
    int xcd_rows;                               // set by the launcher: > 0 = XCD-aware tile order over this many row blocks
    int persistent;                             // set by the launcher (gemm256): workgroups walk tiles slot, slot + grid / 8, ... of their XCD's list
    // ragged batches: segment s holds seg_m[s] output rows (device int32 (nseg); NULL: M for all); seg_m_host = the same on the host
    const int32_t* seg_m;
    const int32_t* seg_m_host;
    // gemmcu.hip's transposed-convolution mode (si_launch_gemmcu_tc; fp16 operands and output): tap t of row m reads input row
    // m + t * tc_dil of a segment of tc_rows_in rows (zeros outside), leaky-ReLU(tc_slope) on the input, output element (m, n) at
    // m * ldo + n + tc_ooff of its segment, dropped outside [0, tc_olimit)
    int tc_dil, tc_rows_in; float tc_slope; long tc_ooff, tc_olimit;
    const int32_t* tc_seg_lin; const int32_t* tc_seg_orows; int tc_olim_mul;   // ragged batches: segment s reads tc_seg_lin[s] rows, keeps [0, tc_seg_orows[s] * tc_olim_mul)
};
int si_launch_lingemm(si_ctx* ctx, const LinGemmParams& p, hipStream_t st);
// The same contract on 256 x 256 tiles with LDS-DMA staging (gemm256.hip), for the shapes whose tiles fill the chip; returns 1 otherwise.
int si_launch_gemm256(si_ctx* ctx, const LinGemmParams& p, hipStream_t st);
int si_opt_gemm256(const si_ctx* ctx);      // SI_ENC_GEMM256: 0 never, 1 by the shape rule (default), 2 whenever the shape allows

// The same contract as ONE tile per CU (gemmcu.hip: 16 waves, tile shape per instantiation), for the flat M = B * T GEMMs of the
// transformer whose tiles then number at most the CUs; returns 1 otherwise.  Bit-identical to the other two.
int si_launch_gemmcu(si_ctx* ctx, const LinGemmParams& p, hipStream_t st);
// The generator's early upsamplers on the fp16 stream as the same kernel (TapGemmParams of a ConvTranspose1d in its two-tap form);
// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller uses the tap-GEMM).
int si_launch_gemmcu_tc(si_ctx* ctx, const TapGemmParams& p, hipStream_t st, bool always = false);   // always: also where tiles are mostly padding
bool si_gemmcu_tc_covers(si_ctx* ctx, const TapGemmParams& p, bool always = false);                   // would the call above launch?
int si_opt_gemmcu(const si_ctx* ctx);       // SI_ENC_GEMMCU: 0 never, 1 by the shape rule (default), 2 whenever the shape allows, 10 + c: instantiation c

// ------------------------------------------------------------------------------------------------
// encoder kernels (encoder_kernels.hip)
// ------------------------------------------------------------------------------------------------
struct WaveNormParams {           // A0 + A1 (group-norm flavour)
    const float* wav;             // (B, N) raw
    const int32_t* mask_start;    // (B) or null
    const int32_t* mask_len;      // (B) or null
    int B, N, L1;                 // L1 = conv0 output length
    int C, K, S;                  // conv0 channels / kernel / stride
    int normalize;                // 0: wav is already zero-mean/unit-variance
    const int32_t* valid_len;     // (B) or null: samples of each right-padded clip that are real; the rest is padding
                                  // (statistics over the real samples only, padding reads as 0 AFTER the normalisation:
                                  // transformers/models/wav2vec2/feature_extraction_wav2vec2.py:88-91)
    float norm_eps;               // variance epsilon of the normalisation: 1e-7 (HF processor) / 1e-5 (F.layer_norm(x, x.shape),
                                  // I_da/src/hubert_feature_reader.py:53-54)
    const double* pre_add;        // (B) or null: value added to every sample BEFORE the zero mask, in fp64, rounded to fp32
                                  // once -- `(y + 1e-6) * mask` on the float64 clip (I_da/scripts/inpainting.py:187-192)
    const int32_t* seg_L1;        // (B) or null: ragged batches -- conv0 output rows of each clip's OWN length (with valid_len =
                                  // its samples): GroupNorm statistics over those rows only, rows past them are not written
};

int si_launch_wave_stats(si_ctx* ctx, const WaveNormParams& p, double* stats /*B*2: mean, rstd*/, hipStream_t st);
// conv0 -> GroupNorm(C groups) -> GELU, channels-last out (B, L1, C)
int si_launch_conv0_groupnorm(si_ctx* ctx, const WaveNormParams& p, const double* stats, const float* w /*[C][K]*/,
                              const float* gamma, const float* beta, double* partials, float* affine, float* out,
                              hipStream_t st, unsigned short* out16 = nullptr /* write bf16 there INSTEAD of fp32 into out */);
// conv0 (+bias) only, channels-last out, for the layer-norm flavour (LN+GELU applied by si_launch_layernorm)
int si_launch_conv0_affine(si_ctx* ctx, const WaveNormParams& p, const double* stats, const float* w, const float* bias,
                           float* affine, float* out, hipStream_t st);
size_t si_conv0_partials_bytes(int B, int N);

// y = LN(x [+ add]) * gamma + beta over the last dim C (rows x C), optional GELU afterwards
// y16 (optional): bf16 copy of y for a bf16 GEMM consumer; y may then be NULL (bf16 output only)
int si_launch_layernorm(si_ctx* ctx, const float* x, const float* add, const float* gamma, const float* beta, float* y,
                        long rows, int C, float eps, int gelu, hipStream_t st, unsigned short* y16 = nullptr, float* stats = nullptr);

// softmax(q k^T / sqrt(64)) v for head_dim 64; qkv (B, T, 3H) packed [q | k | v]; out (B, T, H)
// out16 (optional): write the result as bf16 there INSTEAD of fp32 into out
// bf16_products: with out16 given, run both products on bf16 MFMA (fp32 softmax); false keeps the exact-fp32 kernel
// valid_frames (B) or null: keys >= valid_frames[b] are padding and excluded for every query (modeling_hubert.py:250-251)
// row_off (B + 1, device) or null: ragged batches on PACKED rows -- clip b = rows [row_off[b], row_off[b + 1]); T is then the
// longest clip's frame count (grid sizing) and real_t2 = sum of T_b^2 (the algorithmic flop count)
int si_launch_attention(si_ctx* ctx, const float* qkv, float* out, int B, int T, int H, int heads, hipStream_t st,
                        unsigned short* out16 = nullptr, bool bf16_products = true, const int32_t* valid_frames = nullptr,
                        const int32_t* row_off = nullptr, double real_t2 = 0.0);
// bf16 encoder mode: q | k | v arrive as bf16 (B, T, 3H) from the QKV GEMM's epilogue; both products on bf16 MFMA
int si_launch_attention_bf16in(si_ctx* ctx, const unsigned short* qkv16, int B, int T, int H, int heads, hipStream_t st,
                               unsigned short* out16, const int32_t* valid_frames = nullptr, const int32_t* row_off = nullptr,
                               double real_t2 = 0.0);
// HuBERT's positional conv in the bf16 encoder mode (posconv.hip): out = x + gelu(conv(x) + b) on (rows, H) fp32, the group's weights
// streamed once per workgroup; returns 1 when the shape is not covered (the caller runs the tap-GEMM)
int si_launch_posconv(si_ctx* ctx, const float* x, float* out, const void* w, const float* bias, int B, int T, int Tmax, int H, int groups,
                      int ntaps, int Npad, int pad, const int32_t* row_off, const int32_t* lens, double rows_total, hipStream_t st);
// ragged batches: (B, Tmax, C) padded rows <-> packed rows [row_off[b], row_off[b + 1]); unpack zeroes the padded rows
int si_launch_repack_rows(si_ctx* ctx, const float* src, float* dst, int B, int Tmax, int C, const int32_t* row_off, bool unpack, hipStream_t st);
// valid_frames[b] = conv-stack length of valid_len[b] samples (modeling_hubert.py:664-677), clamped to [1, T]
int si_launch_frame_lengths(si_ctx* ctx, const int32_t* valid_len, int B, int nconv, const int32_t* kernels, const int32_t* strides,
                            int T, int32_t* valid_frames, hipStream_t st);
// x[b][t][:] = 0 for t >= valid_frames[b] (padded frames of the projected states, modeling_hubert.py:428-431)
int si_launch_zero_padded_rows(si_ctx* ctx, float* x, int B, int T, int H, const int32_t* valid_frames, hipStream_t st);

// cosine arg-max against centred centroids + splice of the raw centroid into mel (A10..A13)
int si_launch_codebook_splice(si_ctx* ctx, const float* feats, int B, int T, int D, const int32_t* frame_pos, int Lm,
                              const float* cb_centered /*K x D*/, const float* cb_raw /*K x D*/,
                              const float* cb_rnorm /*K*/, int K, float* mel, int Tm, int64_t* labels, hipStream_t st,
                              const int32_t* frame_cnt = nullptr /* (B): frames replaced per clip (ragged batches), <= Lm */);

// out[b][t] = (t < first[b] || t >= last[b]) ? clean[b][t] : masked[b][t]  (I_da/scripts/inpainting.py:209-214)
int si_launch_code_splice(si_ctx* ctx, const int64_t* clean, const int64_t* masked, const int32_t* first, const int32_t* last, int B, int T,
                          int64_t* out, hipStream_t st);

// mel[b, :, pos_b + j] = cb_raw[labels[b, j]] (the expected_inpaint splice, I_ea/predict.py:177-189)
int si_launch_codebook_gather(si_ctx* ctx, const int64_t* labels, int B, int D, const int32_t* frame_pos, int Lm,
                              const float* cb_raw, int K, float* mel, int Tm, hipStream_t st);

// k-means unit assignment: labels[row] = argmin_k ||x_row - c_k||^2 (first minimum); dist (optional) = that squared distance
// cnorm_scratch (K floats, device) or null: with it (and D % 32 == 0) the distance GEMM runs on the matrix pipe (exact-fp32 MFMA)
int si_launch_kmeans_assign(si_ctx* ctx, const float* x, long rows, int D, const float* cent, int K, int64_t* labels, float* dist,
                            hipStream_t st, float* cnorm_scratch = nullptr);

// loss half of LossFunction.cos_sim + cos_sim_target_labels (f-4): per-frame terms 1 - cos(v, c_target), their
// fixed-order sum, arg-max labels and cos(c_pred, c_target)
int si_launch_codebook_metrics(si_ctx* ctx, const float* feats, int B, int T, int D, const int32_t* frame_pos, int Lm,
                               const float* cb_centered, const float* cb_rnorm, int K, const int64_t* target, float* terms,
                               float* loss, int64_t* pred, float* cos_pt, hipStream_t st);

// mel-domain and waveform metrics (metrics_kernels.hip; I_ea/metrics.py:38-62,127-142)
int si_launch_mel_metrics(si_ctx* ctx, const float* a, const float* b, int B, int D, int L, const float* center, float* out, hipStream_t st);
int si_launch_sisdr(si_ctx* ctx, const float* est, const float* ref, int B, int n, float* out, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// mel front-end kernels (frontend_kernels.hip)
// ------------------------------------------------------------------------------------------------
// n_len / tm_len (B, device) or null: ragged batches -- clip b holds n_len[b] samples (row stride N) and tm_len[b] frames (stride Tm)
int si_launch_wave_peak(si_ctx* ctx, const float* wav, const int32_t* ms, const int32_t* me, int B, int N, float* peak,
                        hipStream_t st, const int32_t* n_len = nullptr);
// mask -> normalise*0.95 -> reflect-pad -> Hann window, as the (B*Tm, kc + nfft / 2) matrix of FOLDED frames
// [w[0], w[k] + w[nfft - k] (k = 1 .. nfft/2 - 1), w[nfft/2], zeros up to kc | 0, w[k] - w[nfft - k]]: the operands of the two
// half-size DFT GEMMs (cosine / sine part)
int si_launch_mel_frames(si_ctx* ctx, const float* wav, const int32_t* ms, const int32_t* me, const float* peak,
                         const float* hann, int B, int N, int Tm, int hop, int pad, int nfft, int kc, int normalize, float* frames,
                         hipStream_t st, const int32_t* n_len = nullptr, const int32_t* tm_len = nullptr);
// spec rows [re | pad | im at im_off] -> sqrt(re^2+im^2+1e-9) -> banded mel basis -> log(clamp 1e-5) -> mel (B, nmel, Tm)
int si_launch_mel_project(si_ctx* ctx, const float* spec, int ld_spec, int nbin, int im_off, const float* basis_t, const int32_t* lo,
                          const int32_t* hi, int nmel, int B, int Tm, float* mel, hipStream_t st, const int32_t* tm_len = nullptr);

// polyphase FIR resampler (upfirdn with resample_poly's centring); taps are device fp32, already pre-padded
int si_launch_resample_poly(si_ctx* ctx, const float* x, int B, int n_in, const float* taps, int ntaps, int up, int down,
                            int pre_remove, int n_out, float* y, hipStream_t st);

// resampy / librosa 0.9.1 `kaiser_best` band-limited interpolation (frontend_kernels.hip); tables are device float64
int si_launch_resample_sinc(si_ctx* ctx, const float* x, const int32_t* n_len, int B, int n_in, const double* win, const double* dwin, int nwin,
                            int num_table, int step, double scale, double ratio, const double* time_reg, int n_out, float* y, hipStream_t st);
// `audio * 32768` + truncating int16 cast (I_ea/predict.py:204-206)
int si_launch_pcm16(si_ctx* ctx, const float* wav, long n, int16_t* out, hipStream_t st);

// erf-GELU of the bf16 encoder's GEMM epilogues (lingemm.hip, gemm256.hip: the SAME function, their results are bit-identical).
// erf by Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 -- below one fp32 ulp of the result for |x| >= 1 and far below the bf16
// rounding every consumer of these outputs applies; 15 VALU operations instead of libm erff's ~31 with two divergent branches.
__device__ __forceinline__ float si_gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = __builtin_fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    const float erf = __builtin_copysignf(fmaf(-poly, e, 1.0f), z);
    return 0.5f * x * (1.0f + erf);
}

// LayerNorm applied to one element: the ONE expression of it on the path.  layernorm_kernel writes y with it; a GEMM epilogue whose
// residual is a LayerNorm output that was never stored (LinGemmParams::res_stats) recomputes the same value from the same floats.
__device__ __forceinline__ float si_ln_apply(float x, float mean, float rstd, float g, float b) { return fmaf((x - mean) * rstd, g, b); }

// leaky-ReLU(0.1) of an accumulator as max(v, 0.1 v) in two VALU ops: fmaxf() would first canonicalise both operands
// (a v_max_f32 v, v, v each), which nothing downstream of an MFMA accumulator needs
__device__ __forceinline__ float si_lrelu01(float v) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(0.1f * v));
    return r;
}

// lrelu(0.1) -> ConvTranspose1d(Cin -> Cin / 2, k = 4, stride 2) of the generator's late stages on the fp16 stream, as a streaming
// GEMM over overlapping input rows (upsample.hip): GEMM row m = input row m, columns n = phase * Cout + co, taps read rows m, m - 1.
struct UpsampleParams {
    const unsigned short* x16;        // (B, Lin, Cin) raw fp16
    const unsigned short* w;          // [taps][N][Cin] fp16 (the tap-GEMM's packing)
    const float* bias;                // [N]: the layer's bias repeated per phase
    unsigned short* out16;            // (B, Lout, Cout) raw fp16
    int B, Lin, M;                    // M GEMM rows per clip (Lin + 1 for k = 4, stride 2, padding 1)
    int Cin, N, taps;
    long ooff;                        // GEMM element (m, n) is element m * N + n - ooff of its clip's output (ooff = padding * Cout)
    long o_clip_stride, o_clip_elems; // elements between clips / per clip of the output
    // ragged batches (NULL: uniform): clip b holds lens_lin[b] input rows, lens_m[b] GEMM rows and lens_lin[b] * N output elements
    // (k = 2 * stride: N output elements per input row); *_host: the same on the host; Lin / M / o_clip_elems = the longest clip's
    const int32_t* lens_lin; const int32_t* lens_m; const int32_t* lens_m_host;
    int total_tiles;                  // set by the launcher
};
// SI_OK when launched, negative on error, 1 when the shape is not covered (the caller runs the tap-GEMM)
int si_launch_upsample_stream(si_ctx* ctx, const UpsampleParams& p, hipStream_t st);

// One ResBlock1 step y' = (y + conv2(lrelu(conv1(lrelu(y)) + b1)) + b2) * alpha [+ previous y'] as one kernel on the raw
// fp16 activation stream (respair.hip: C = 32 / 64; respair_wide.hip: C = 128 / 256).  Returns 1 when the shape is not covered.
struct ResPairParams {
    const unsigned short* y16;   // [B][L][C] raw fp16 activation stream (input and residual)
    unsigned short* out16;       // [B][L][C] raw fp16
    const unsigned short* w1;    // [k][C][C] fp16 (tap, n, ci)
    const unsigned short* w2;
    const float* b1;
    const float* b2;
    int B, L, k, dil;
    float alpha;                 // out = (conv2 + b2 + y) * alpha
    int accumulate;              // out += previous out16
    // ragged batches (NULL: every clip holds L rows): clip b holds lens[b] rows at stride L; lens_host = the same on the host
    const int32_t* lens; const int32_t* lens_host;
    int total_tiles;             // set by the launcher
    float out_slope;             // 0 or 1: none; otherwise out = leaky_relu(out, out_slope) before the fp16 rounding -- the activation of the
                                 // only consumer (the next stage's upsampler on gemmcu.hip), applied by the producer (respair_wide.hip only)
};
int si_launch_respair_wide(si_ctx* ctx, int C, const ResPairParams& p, hipStream_t st);
// A whole ResBlock1 -- three (c1, c2) pairs chained, the residual stream kept in LDS -- as one kernel (reschain.hip: C = 32).
// out = x_3 * alpha [+ previous out]; returns 1 when the shape is not covered.
struct ResChainParams {
    const unsigned short* y16;   // [B][L][C] raw fp16 activation stream
    unsigned short* out16;       // [B][L][C] raw fp16
    const unsigned short* w1[3]; // per pair: [k][C][C] fp16 (tap, n, ci)
    const unsigned short* w2[3];
    const float* b1[3];
    const float* b2[3];
    int B, L, k;
    int dil[3];                  // dilation of c1 per pair (c2 has dilation 1)
    float alpha;
    int accumulate;
    const int32_t* lens; const int32_t* lens_host;   // ragged batches: as ResPairParams
    int total_tiles;
};
int si_launch_reschain(si_ctx* ctx, int C, const ResChainParams& p, hipStream_t st);
int si_launch_respair(si_ctx* ctx, int C, const unsigned short* y16, unsigned short* out16, const void* w1, const void* w2,
                      const float* b1, const float* b2, int B, int L, int k, int dil, float alpha, int accumulate, hipStream_t st,
                      const int32_t* lens = nullptr, const int32_t* lens_host = nullptr, float out_slope = 1.f);

// ------------------------------------------------------------------------------------------------
// vocoder kernels (vocoder_kernels.hip)
// ------------------------------------------------------------------------------------------------
// mel (B, D, Tm) channels-first -> (B, Tout, ldo) channels-last, time-stretched (stretch=1) or copied;
// channels D..ldo-1 are written as zero.
// tm_len / tout_len (B, device) or null: ragged batches -- clip b holds tm_len[b] mel frames (stride Tm) and tout_len[b] output rows (stride Tout)
int si_launch_extend_mel(si_ctx* ctx, const float* mel, int B, int D, int Tm, int Tout, int stretch, float* out, int ldo,
                         hipStream_t st, const int32_t* tm_len = nullptr, const int32_t* tout_len = nullptr);
int si_launch_extend_mel_cf(si_ctx* ctx, const float* mel, int B, int D, int Tm, int Tout, float* out, hipStream_t st);   // (B, D, Tm) -> (B, D, Tout), channels-first
// I_da CodeGenerator front (f-2): embedding look-ups + frame repeat + channel concat -> (B, nparts * E, F) channels-first
int si_launch_small_conv1d(si_ctx* ctx, const float* x, const float* w, const float* bias, const float* res, float* y, int B, int Cin,
                           int Tin, int Cout, int Tout, int K, int stride, int dil, int pad, int relu_in, int channels_last, hipStream_t st);
// the whole F0 encoder in one launch, activations in LDS (vocoder_kernels.hip); 1 = does not fit, launch layer by layer
int si_launch_f0enc_fused(si_ctx* ctx, const float* weights, const float* f0, int B, int T, float* h_out, int in_width, int out_width, int width,
                          int n_state, int depth, int down_t, int stride_t, int growth, int dk, int dpad, double macs, hipStream_t st);
int si_launch_unit_frontend(si_ctx* ctx, const int64_t* code, int Fc, const int64_t* f0_code, int Fp, const float* spk_emb,
                            const float* emb_c, int Kc, const float* emb_p, int Kp, int E, int B, float* out, hipStream_t st);
// leaky_relu(0.01) -> Conv1d(C -> 1, k, pad k/2) -> tanh ; x (B, L, C) channels-last -> wav (B, L)
// lens / lens_host (B) or null: ragged batches -- clip b holds lens[b] rows (stride L) and as many output samples (stride L)
int si_launch_conv_post(si_ctx* ctx, const float* x, const float* w /*[k][C]*/, const float* bias, int B, int L, int C, int k,
                        float* wav, hipStream_t st, const unsigned short* x16 = nullptr /* raw fp16 input instead of x */,
                        const int32_t* lens = nullptr, const int32_t* lens_host = nullptr);
