/*
 * si_hip.h -- C ABI of libsi_hip.so: the MI355X (gfx950) implementation of the I_ea predict hot path.
 *
 * The reference (Fireflies-17/Speech-Inpainting) has no FFI; its seam is three Python module calls.
 * Each entry point below replaces one of them (paths relative to /root/reference):
 *
 *   si_hubert_forward    <- CustomModel.forward                I_ea/model.py:80-89 (called at I_ea/predict.py:163)
 *                           + the zero-mask + processor normalise in front of it (I_ea/predict.py:132-141)
 *   si_codebook_splice   <- frame gather + LossFunction.cos_sim arg-max + centroid splice
 *                           I_ea/predict.py:164-168,171,184-187 ; I_ea/loss_fn.py:44-47
 *   si_codebook_metrics  <- LossFunction.cos_sim loss + cos_sim_target_labels (row f-4)
 *                           I_ea/loss_fn.py:29-62 ; I_ea/predict.py:171-173
 *   si_mel_metrics       <- Metrics.avg_cosine_sim / avg_d2_dist / rmse (row f-4)   I_ea/metrics.py:38-62
 *   si_sisdr             <- Metrics.sisdr (row f-4)                                 I_ea/metrics.py:127-142
 *   si_unit_frontend     <- CodeGenerator.forward's embedding / _upsample / concat front (row f-2)   I_da/src/model.py:79-119,148-189
 *   si_f0_encoder_forward <- FoVQVAE.encoder(fo) inside CodeGenerator.forward (row f-2)   I_da/src/model.py:160-163 ; I_da/src/modules/jukebox.py:11-116,200-262
 *   si_hubert_extract_features <- HubertFeatureReader.get_feats (row f-2)   I_da/src/hubert_feature_reader.py:44-67 (called at
 *                           I_da/scripts/inpainting.py:195-198) + the corruption `(y + 1e-6) * mask` in front of it (:186-192)
 *   si_code_splice       <- the unit splice (row f-2)                        I_da/scripts/inpainting.py:209-214
 *   si_kmeans_assign     <- kmeans_model.predict(feats) (row f-2)   I_da/scripts/inpainting.py:204-205 ;
 *                           ApplyKmeans.__call__                    I_ea/dataset/km_label.py:20-24
 *   si_resample_poly     <- librosa.load(..., sr=22050 / 16000) resampling (row f-3)   I_ea/predict.py:79-80
 *   si_mel_frontend      <- 22.05 kHz masking + normalize*0.95 + get_mel
 *                           I_ea/predict.py:99-106 ; I_ea/dataset/mel_dump.py:40-98
 *   si_hifigan_forward   <- extend_mel + Generator.forward     I_ea/hifi_gan/inference_modified.py:16-19 ;
 *                           I_ea/hifi_gan/models.py:107-123 (called at I_ea/predict.py:189,203)
 *   si_load_weights      <- model.load_state_dict / generator.load_state_dict + remove_weight_norm + ApplyKmeans
 *                           I_ea/predict.py:117-122,149 ; I_ea/hifi_gan/models.py:125-132 ; I_ea/dataset/km_label.py:12-24
 *
 * Conventions: plain C types only; every function returns 0 on success and a negative SI_E* code on
 * failure (message via si_last_error); no exceptions cross the boundary.  A context is bound to one device,
 * is not thread-safe, enqueues all work on the caller's HIP stream and never synchronises it.  All data
 * pointers are DEVICE pointers owned by the caller unless the parameter says "host".
 * Activations are fp32.  Layouts are the reference's: wave (B, N); feats (B, T, D); mel (B, D, Tm)
 * channels-first; waveform (B, Tm' * hop).
 */
#ifndef SI_HIP_H
#define SI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SI_ABI_VERSION 3

enum {
    SI_OK = 0,
    SI_EINVAL = -1,      /* bad argument / unsupported shape */
    SI_ENOMEM = -2,      /* workspace too small or device allocation failed */
    SI_EHIP = -3,        /* a HIP runtime call failed */
    SI_ESTATE = -4,      /* call order violated (e.g. forward before weights) */
    SI_EWEIGHTS = -5     /* missing / mis-shaped tensor in the checkpoint index */
};

/* arithmetic of the contraction (accumulation is always fp32) */
enum {
    SI_MATH_F32 = 0,     /* v_mfma_f32_32x32x2_f32: exact fp32 fma chain */
    SI_MATH_BF16 = 1,    /* v_mfma_f32_32x32x16_bf16, operands rounded to bf16 */
    SI_MATH_BF16X3 = 2,  /* hi/lo split bf16, 3 MFMAs per product: ~2^-16 relative operand error */
    SI_MATH_F16 = 3      /* v_mfma_f32_32x32x16_f16, operands rounded to fp16 (saturating): 2^-12 relative operand error */
};

#define SI_MAX_CONV 8
#define SI_MAX_UPS 8
#define SI_MAX_RB 4
#define SI_MAX_DIL 4

typedef struct si_ctx si_ctx;
typedef void* si_stream_t;      /* hipStream_t */

/* Architecture + arithmetic of one model pair.  HuBERT fields follow the HuggingFace config.json
 * (I_ea/dataset/config.json:62-124), vocoder fields the HiFi-GAN json (I_ea/hifi_gan/config_v1.json). */
typedef struct si_model_desc {
    int32_t struct_size;            /* = sizeof(si_model_desc), ABI check */
    /* encoder */
    int32_t hidden_size, num_layers, num_heads, intermediate_size;
    int32_t num_conv;
    int32_t conv_dim[SI_MAX_CONV], conv_kernel[SI_MAX_CONV], conv_stride[SI_MAX_CONV];
    int32_t conv_bias;              /* 0 base / 1 large */
    int32_t feat_norm_layer;        /* 0 = "group" (GroupNorm on conv0 only), 1 = "layer" (LayerNorm on every conv) */
    int32_t stable_layer_norm;      /* 0 post-LN (base), 1 pre-LN (large) */
    int32_t pos_conv_kernel, pos_conv_groups;
    int32_t feat_proj_layer_norm;
    float   layer_norm_eps;
    int32_t codebook_dim;           /* final_layers Linear(H -> codebook_dim), 80 */
    int32_t num_clusters;           /* K centroids, 100 or 500 */
    /* vocoder */
    int32_t num_mels;
    int32_t num_ups;
    int32_t up_rates[SI_MAX_UPS], up_kernels[SI_MAX_UPS];
    int32_t up_initial_channel;
    int32_t num_rb;                 /* resblocks per stage (3) */
    int32_t rb_kernels[SI_MAX_RB];
    int32_t num_dil;                /* dilations per resblock (3) */
    int32_t rb_dilations[SI_MAX_RB][SI_MAX_DIL];
    /* arithmetic */
    int32_t encoder_math;           /* SI_MATH_* for the encoder GEMMs/convs (attention + head stay fp32) */
    int32_t vocoder_math;           /* SI_MATH_* for the generator convs */
    int32_t vocoder_chunk;          /* clips per vocoder pass (0 = library default; sized to the Infinity Cache) */
    int32_t resblock_type;          /* 1 (or 0): ResBlock1 -- num_dil (conv_dilated, conv) pairs per block (config_v1 / v2, I_da);
                                       2: ResBlock2 -- num_dil single dilated convs per block, x = x + conv(lrelu(x))
                                       (I_ea/hifi_gan/models.py:52-73, selected at :89 by config_v3.json:2) */
} si_model_desc;

int si_version(void);

/* Create a context on `device_id`.  No device memory is allocated until weights are loaded. */
int si_create(si_ctx** out, int device_id, const si_model_desc* desc);
void si_destroy(si_ctx* ctx);

/* Last error text of this context (or of si_create when ctx == NULL).  Never NULL. */
const char* si_last_error(const si_ctx* ctx);

/* Load a checkpoint.  `host_blob` (host memory) holds fp32 tensors; `index` is text, one tensor per line:
 *     <name> <byte_offset> <ndim> <d0> ... <d(ndim-1)>
 * Names are the reference's state-dict keys: "base_model.<hf key>", "final_layers.{0,1}.{weight,bias}",
 * "generator.<Generator key>" (folded ".weight" or weight-norm ".weight_g"/".weight_v"; ResBlock1 blocks hold
 * "convs1.<n>" / "convs2.<n>", ResBlock2 blocks "convs.<n>"), "codebook" (K, D).
 * Weight-norm is folded here (dim 0 for the generator, dim 2 for the positional conv, in either the
 * "parametrizations.weight.original{0,1}" or the legacy "weight_g/weight_v" spelling).  The tensors are
 * re-laid-out for the kernels into one packed device blob owned by the context. */
int si_load_weights(si_ctx* ctx, const void* host_blob, size_t nbytes, const char* index);

/* Multi-GPU: ranks that do not read the checkpoint allocate the (identically laid out) packed blob with
 * si_alloc_weights and receive its bytes by an RCCL broadcast into si_weights_device_ptr. */
int si_alloc_weights(si_ctx* ctx);
int si_weights_device_ptr(si_ctx* ctx, void** ptr, size_t* nbytes);
/* The blob starts with a fingerprint of the layout it was packed for (model desc + the context's arithmetic-path options).
 * si_weights_check compares it with THIS context's plan (one small synchronous device-to-host copy): SI_EWEIGHTS when the
 * source rank planned a different layout or the bytes have not arrived.  Call it after the broadcast; the forwards also
 * run it once before the first use of a blob that came from si_alloc_weights. */
int si_weights_check(si_ctx* ctx);

/* Workspace (device scratch) needed for a batch of B clips of N samples and Tm mel frames. */
int si_workspace_bytes(si_ctx* ctx, int B, int N, int Tm, size_t* out);

/* Encoder: 16 kHz clips -> (B, T, codebook_dim).  Samples [mask_start[b], mask_start[b]+mask_len[b])
 * are zeroed, then (normalize != 0) each clip is normalised to zero mean / unit variance (eps 1e-7); both are
 * fused into the load of the first conv.  normalize = 0 takes `wav` as the processor's input_values
 * (already normalised), which is what CustomModel.forward receives in the reference.
 * mask_start / mask_len: device int32 (B), may be NULL (no mask).
 * T = the conv-stack length of N (modeling_hubert.py:664-677). */
int si_hubert_forward(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, int normalize,
                      int B, int N, float* out_feats, void* workspace, size_t workspace_bytes, si_stream_t stream);

/* The same for RIGHT-PADDED batches -- `CustomModel.forward(input_values, attention_mask)` with a mask that is not all ones
 * (I_ea/model.py:80-85 -> modeling_hubert.py:921-932): valid_len (device int32 (B), samples) marks the real part of each clip.
 * normalize != 0 reproduces the processor's padded normalisation (statistics over the real samples, padding = 0 after it);
 * the feature extractor and the projection run over the whole padded input (HuBERT-base's GroupNorm therefore sees the
 * padding, exactly as the reference's does), the padded frames of the projected states are zeroed and excluded as
 * attention keys (:428-437, frame lengths by :664-689).  Outputs are defined on ALL T frames, as in the reference.
 * valid_len = NULL is si_hubert_forward. */
int si_hubert_forward_padded(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len,
                             const int32_t* valid_len, int normalize, int B, int N, float* out_feats, void* workspace,
                             size_t workspace_bytes, si_stream_t stream);

/* RAGGED batches (BASELINE configs[4]: clips of different lengths, I_ea/config.yaml:11 "10.1 s", I_ea/mask_pos_len.py:28-35): B clips
 * of DIFFERENT lengths in one call, each clip's result EQUAL to that clip run alone -- the reference's script handles one file of
 * any length per invocation (I_ea/predict.py:76-207) -- and not the padded semantics of si_hubert_forward_padded (whose GroupNorm
 * sees the padding, as HuggingFace's does).  wav (B, N): clip b occupies the first sample_len[b] samples of its row, the rest is
 * ignored.  sample_len: HOST int32 (B) (launch grids depend on it).  mask_start / mask_len as si_hubert_forward.
 * Per clip: the processor's statistics over its own samples, conv0's GroupNorm statistics over its own conv rows, every strided
 * convolution stops at its own length (modeling_hubert.py:664-677), the transformer runs on the packed rows of all clips (the
 * positional conv's zero padding begins at the clip's own last frame, attention sees its own frames only).
 * out_feats (B, T, codebook_dim) with T = si_num_frames(N): rows >= si_num_frames(sample_len[b]) of clip b are zero. */
int si_hubert_forward_varlen(si_ctx* ctx, const float* wav, const int32_t* mask_start, const int32_t* mask_len, const int32_t* sample_len,
                             int normalize, int B, int N, float* out_feats, void* workspace, size_t workspace_bytes, si_stream_t stream);

/* I_da's encoder call: `HubertFeatureReader.get_feats` (I_da/src/hubert_feature_reader.py:44-67) =
 * optional `F.layer_norm(x, x.shape)` over the whole clip (:53-54, when the checkpoint's task.cfg.normalize is set) followed by
 * fairseq's `model.extract_features(source, padding_mask=None, mask=False, output_layer=L)` (:60-65): the feature extractor,
 * LayerNorm + projection, positional conv, then transformer layers 0..L-1 ONLY; the output is layer L-1's output (B, T, H) --
 * in the pre-LN ("layer_norm_first") flavour the residual stream WITHOUT the encoder's final LayerNorm, which fairseq
 * applies only when no layer is requested.  No head, no codebook.
 * fairseq is not in the build image and the reference pins no version of it; the arithmetic is the HuBERT architecture
 * this library already implements for I_ea (the transformers port of the same checkpoints), so the entry is pinned
 * against `HubertModel(..., output_hidden_states=True).hidden_states[L]` (tests/golden/hidden_layers.npz).
 * mask_start / mask_len (device int32 (B), samples, or NULL) and pre_mask_add (device fp64 (B) or NULL) restate the
 * script's corruption `y_inpainting = (y + 1e-6) * mask` on the float64 clip (I_da/scripts/inpainting.py:186-192): the
 * addend is applied in fp64 to every sample of clip b and rounded to fp32 once, then the span is zeroed -- fused into
 * the first conv's loaders like I_ea's mask.  Clips of the clean stream pass mask_len = 0 and pre_mask_add = 0. */
typedef struct si_extract_desc {
    int32_t struct_size;        /* = sizeof(si_extract_desc) */
    int32_t output_layer;       /* L in 1..num_layers */
    int32_t normalize;          /* 0: wav is used as is; 1: HF processor (x - mean) / sqrt(var + 1e-7) (I_ea);
                                   2: F.layer_norm(x, x.shape): (x - mean) / sqrt(var + 1e-5) (I_da, task.cfg.normalize) */
    int32_t reserved;
} si_extract_desc;
int si_hubert_extract_features(si_ctx* ctx, const si_extract_desc* x, const float* wav, const int32_t* mask_start,
                               const int32_t* mask_len, const double* pre_mask_add, int B, int N, float* out_hidden,
                               void* workspace, size_t workspace_bytes, si_stream_t stream);

/* I_da's code splice (I_da/scripts/inpainting.py:209-214): the units predicted from the corrupted clip survive only
 * inside the mask -- `code_inpainting[: fs // hop] = code[: fs // hop]; code_inpainting[(fs + ms) // hop :] = code[...]`:
 * out[b][t] = (t < first[b] || t >= last[b]) ? code_clean[b][t] : code_masked[b][t], first = frame_start // code_hop_size,
 * last = (frame_start + mask_size) // code_hop_size.  code_* / out device int64 (B, T) (out may alias code_masked);
 * first / last device int32 (B).  Needs no weights. */
int si_code_splice(si_ctx* ctx, const int64_t* code_clean, const int64_t* code_masked, const int32_t* first, const int32_t* last,
                   int B, int T, int64_t* out, si_stream_t stream);

/* Codeword decision + splice: for b, j < Lm:  label = argmax_k cos(feats[b, pos_b + j], C_k - mean(C));
 * mel[b, :, pos_b + j] = C_label.   feats (B, T, D); frame_pos device int32 (B); mel (B, D, Tm) in/out;
 * labels device int64 (B, Lm), may be NULL. */
int si_codebook_splice(si_ctx* ctx, const float* feats, int B, int T, const int32_t* frame_pos, int Lm,
                       float* mel, int Tm, int64_t* labels, si_stream_t stream);

/* The same with a per-clip frame COUNT (ragged batches; blind inpainting replaces all of a clip's own frames): frame_cnt device
 * int32 (B), 0 <= frame_cnt[b] <= Lm; labels (B, Lm) get -1 past a clip's count. */
int si_codebook_splice_varlen(si_ctx* ctx, const float* feats, int B, int T, const int32_t* frame_pos, const int32_t* frame_cnt, int Lm,
                              float* mel, int Tm, int64_t* labels, si_stream_t stream);

/* Splice GIVEN codewords: mel[b, :, pos_b + j] = C[labels[b, j]] -- the script's `expected_inpaint` branch, which puts the
 * ground-truth centroids where si_codebook_splice puts the predicted ones (I_ea/predict.py:177-189: all_embeds_t_c[0, labels]
 * + center_).  labels device int64 (B, Lm); a label outside [0, K) or a frame outside [0, Tm) leaves that column untouched. */
int si_codebook_splice_labels(si_ctx* ctx, const int64_t* labels, int B, const int32_t* frame_pos, int Lm, float* mel, int Tm,
                              si_stream_t stream);

/* Loss half of the same LossFunction call: `loss, pred = cos_sim(values, labels)` and
 * `cos_sim_target_labels(pred, labels)` (I_ea/loss_fn.py:29-62, called at I_ea/predict.py:171-173).  feats / frame_pos / Lm
 * as si_codebook_splice (pass the gathered values with T = Lm and frame_pos = 0 to mirror the call exactly).
 * target_labels: device int64 (B, Lm).  Outputs (device): loss_terms fp32 (B, Lm) = 1 - cos(v, centred target centroid);
 * loss fp32 (1) = their sum in a fixed order; pred_labels int64 (B, Lm) or NULL; cos_pred_target fp32 (B, Lm) =
 * cos(centred predicted centroid, centred target centroid).  A target outside [0, K) gives NaN for that frame. */
int si_codebook_metrics(si_ctx* ctx, const float* feats, int B, int T, const int32_t* frame_pos, int Lm,
                        const int64_t* target_labels, float* loss_terms, float* loss, int64_t* pred_labels,
                        float* cos_pred_target, si_stream_t stream);

/* k-means unit assignment: labels[r] = argmin_k ||feats[r] - centroids[k]||^2, first minimum on
 * ties -- what `kmeans_model.predict(feats)` returns at I_da/scripts/inpainting.py:204-205 (sklearn minimises
 * ||c||^2 - 2 x.c) and `ApplyKmeans.__call__` at I_ea/dataset/km_label.py:20-24.  feats device fp32 (rows, D),
 * centroids device fp32 (K, D) (caller-owned: I_da's unit codebook lives in HuBERT feature space, not in the context),
 * labels device int64 (rows), sq_dist optional device fp32 (rows).  Needs no weights. */
int si_kmeans_assign(si_ctx* ctx, const float* feats, int64_t rows, int D, const float* centroids, int K, int64_t* labels,
                     float* sq_dist, si_stream_t stream);

/* Mel-domain metrics (row f-4, `Metrics.avg_cosine_sim / avg_d2_dist / rmse`, I_ea/metrics.py:38-62) of two mel segments
 * per clip: mel_a, mel_b device fp32 (B, D, L) channels-first; center device fp32 (D) (the codebook mean the reference
 * subtracts before the cosine) or NULL.  out3 device fp32 (B, 3) = {mean over frames of cos over bins of the centred
 * frames; mean over frames of 20/ln10 * sqrt(mean over bins of ((a - mean_bins a) - (b - mean_bins b))^2); the same with one
 * global mean}.  Needs no weights. */
int si_mel_metrics(si_ctx* ctx, const float* mel_a, const float* mel_b, int B, int D, int L, const float* center, float* out3,
                   si_stream_t stream);

/* Scale-invariant SDR in dB (`Metrics.sisdr`, I_ea/metrics.py:127-142) of est against ref, device fp32 (B, n) each;
 * out device fp32 (B).  eps = float32 machine epsilon, as np.finfo(x_est.dtype).eps for float32 waveforms. */
int si_sisdr(si_ctx* ctx, const float* est, const float* ref, int B, int n, float* out, si_stream_t stream);

/* I_da `CodeGenerator` front (I_da/src/model.py:148-189 in the look-up-table configuration of
 * I_da/configs/LJSpeech/hubert_lut.json): out = concat_channels( emb_c[code], emb_p[f0_code], spk_emb ) with the shorter
 * index series repeated frame-wise up to the longer (`_upsample`, :79-119; the lengths must divide) and the speaker
 * embedding vector repeated over all frames.  code device int64 (B, Fc); f0_code device int64 (B, Fp) or NULL (no pitch
 * part; the indices are what the fixed F0 VQ-VAE emits at :163-165 -- its conv encoder is si_f0_encoder_forward, its quantiser
 * si_kmeans_assign's arg-min); spk_emb device fp32 (B, E) or NULL; emb_c (Kc, E) / emb_p (Kp, E) device fp32
 * tables owned by the caller (they live in the CodeGenerator checkpoint, not in this context).  out device fp32
 * (B, nparts * E, max(Fc, Fp)) channels-first = the input of si_hifigan_forward(stretch = 0) for a generator whose
 * num_mels = nparts * E (384 for hubert_lut.json).  An index outside its table yields NaN.  Needs no weights. */
int si_unit_frontend(si_ctx* ctx, const int64_t* code, int Fc, const int64_t* f0_code, int Fp, const float* spk_emb,
                     const float* emb_c, int Kc, const float* emb_p, int Kp, int E, int B, float* out, si_stream_t stream);

/* The conv encoder of the fixed F0 VQ-VAE that `CodeGenerator.forward` runs on the F0 track (SURVEY 8(f) row f-2;
 * I_da/src/model.py:160-163 -> `FoVQVAE.encoder` = I_da/src/modules/jukebox.py `Encoder` :200-262 with one level of
 * `EncoderConvBlock` :11-116 and `Resnet1D` / `ResConv1DBlock` I_da/src/modules/resnet.py:29-97; configuration
 * I_da/configs/LJSpeech/hubert_lut.json:42-52): down_t x [Conv1d(k = 2 s (2 s + 1 for odd s), stride s, pad s / 2 (+ 1)) ->
 * depth x (x + Conv1d_k1(ReLU(Conv1d_k3, dilation growth^j, padding = dilation (ReLU(x)))))] -> Conv1d(width -> out_width, 3, 1, 1).
 * n_state = int(m_conv * width).  The quantiser behind it (`Bottleneck`, vq.py:117-127: arg-min of |x|^2 - 2 x.k + |k|^2)
 * is si_kmeans_assign on this function's output rows; the look-up + concat is si_unit_frontend.
 * weights: device fp32, packed in module order -- per down block: conv weight [width][cin][k], bias [width], then per
 * res block: k3 weight [n_state][width][3], bias, k1 weight [width][n_state][1], bias; last: weight [out_width][width][3],
 * bias (si_f0_encoder_weight_floats values).  f0: device fp32 (B, in_width, T).  h_out: device fp32 (B, T', out_width),
 * CHANNELS-LAST (the (N T, C) rows the bottleneck's `preprocess` builds at vq.py:92-95), T' = si_f0_encoder_frames(T). */
typedef struct si_f0enc_desc {
    int32_t in_width, out_width, width, n_state, depth, down_t, stride_t, dilation_growth;
} si_f0enc_desc;
size_t si_f0_encoder_weight_floats(const si_f0enc_desc* d);
int si_f0_encoder_frames(const si_f0enc_desc* d, int T);
size_t si_f0_encoder_workspace_bytes(const si_f0enc_desc* d, int B, int T);
int si_f0_encoder_forward(si_ctx* ctx, const si_f0enc_desc* d, const float* weights, const float* f0, int B, int T, float* h_out,
                          void* workspace, size_t workspace_bytes, si_stream_t stream);

/* Polyphase FIR resampler: the sample-rate conversions in front of the path, `librosa.load(path,
 * sr=22050)` / `sr=16000` at I_ea/predict.py:79-80.  y = upfirdn(taps, x, up, down)[pre_remove : pre_remove + n_out], i.e.
 * scipy.signal.resample_poly's arithmetic with a caller-designed filter: taps (device fp32) are the low-pass FIR
 * multiplied by `up` and already zero-pre-padded as resample_poly pads them; the host helper that designs them is
 * speech_inpainting_amd/audio.py::design_resampler.  x device fp32 (B, n_in), y device fp32 (B, n_out).  librosa's own
 * resampler (soxr / resampy) is a different filter; it is absent from the build image, so parity is against scipy. */
int si_resample_poly(si_ctx* ctx, const float* x, int B, int n_in, const float* taps, int ntaps, int up, int down,
                     int pre_remove, int n_out, float* y, si_stream_t stream);

/* librosa's OWN resampler: `librosa.load(path, sr=16000)` at I_ea/predict.py:79-80 is, in the pinned
 * librosa==0.9.1 (requirements.txt:3), `resample(..., res_type='kaiser_best')` = resampy's band-limited interpolation with its
 * `kaiser_best` table.  resampy is a third-party dependency absent from the build image (no version pinned by the reference); its
 * published algorithm is restated (oracle/ref_cpu.py::resample_kaiser_best) and pinned bit for bit by the fixture the reference
 * holds: I_ea/hifi_gan/test_files/LJ001-0001_{22k,16k}.wav (tests/golden/lj001_resample.npz).
 * The caller designs the tables on the host (speech_inpainting_amd/audio.py::design_kaiser_best) and owns them on the device:
 * win / dwin float64 (nwin): the windowed sinc's right wing (scaled by the ratio when down-sampling) and its forward differences;
 * time_reg float64 (n_time): output sample t's input time, 1 / ratio accumulated by repeated addition as resampy's loop does.
 * x (B, n_in) fp32; n_len device int32 (B) or NULL: ragged batches, clip b holds n_len[b] samples; y (B, n_out) fp32: samples
 * >= int(len_b * ratio) are zero (librosa's fix_length padding up to ceil(len * ratio)). */
typedef struct si_sinc_filter {
    int32_t struct_size;        /* = sizeof(si_sinc_filter) */
    int32_t nwin, num_table, step, n_time;
    int32_t reserved;
    double scale, ratio;
    const double* win;
    const double* dwin;
    const double* time_reg;
} si_sinc_filter;
int si_resample_sinc(si_ctx* ctx, const float* x, const int32_t* n_len, int B, int n_in, const si_sinc_filter* f, int n_out, float* y,
                     si_stream_t stream);

/* B6 on the device (I_ea/predict.py:204-206: `audio * 32768`, `.astype('int16')`): out[i] = the fp32 product truncated toward
 * zero; 32768.0 (an exactly saturated +1.0 sample, where the reference's cast is undefined behaviour) gives 32767, NaN gives 0.
 * wav device fp32 (n), out device int16 (n). */
int si_pcm16(si_ctx* ctx, const float* wav, int64_t n, int16_t* out, si_stream_t stream);

/* Vocoder: mel (B, D, Tm) -> time-stretch x441/256 -> generator -> wav_out (B, floor(Tm*441/256) * hop).
 * stretch = 0 skips extend_mel (mel already at the generator's frame rate; output (B, Tm * hop)). */
int si_hifigan_forward(si_ctx* ctx, const float* mel, int B, int Tm, int stretch, float* wav_out,
                       void* workspace, size_t workspace_bytes, si_stream_t stream);

/* `extend_mel` alone (I_ea/hifi_gan/inference_modified.py:16-19): mel (B, num_mels, Tm) -> out (B, num_mels, floor(Tm * 441 / 256)),
 * both channels-first -- the stretch si_hifigan_forward(stretch = 1) applies internally, as a separate step so that a WINDOW of the
 * stretched frames can be vocoded with stretch = 0 (the script's three generator passes differ only around the mask,
 * I_ea/predict.py:123-128,196-207: speech_inpainting_amd/engine.py::vocode_window). */
int si_extend_mel(si_ctx* ctx, const float* mel, int B, int Tm, float* out, si_stream_t stream);

/* Ragged batches: clip b holds mel_len[b] (HOST int32 (B), 1..Tm) frames of its (D, Tm) slab.  The stretch clamps at the clip's
 * own last frame and every convolution's zero padding begins at its own end, so the first si_vocoder_samples(mel_len[b]) samples
 * of row b equal that clip's waveform alone; the rest of the row (rows are si_vocoder_samples(Tm) long) is zero. */
int si_hifigan_forward_varlen(si_ctx* ctx, const float* mel, const int32_t* mel_len, int B, int Tm, int stretch, float* wav_out,
                              void* workspace, size_t workspace_bytes, si_stream_t stream);

/* Log-mel front-end of the vocoder side.  wave22: device fp32 (B, N22), the RAW 22.05 kHz clip.  Per clip the span
 * [mask_start[b], mask_end[b]) (device int32, samples; both NULL = no masking) is zeroed, the clip is divided by its
 * max |x| and scaled by 0.95 (normalize != 0; I_ea/predict.py:99-104), then get_mel (I_ea/dataset/mel_dump.py:40-98,
 * constants :11-20: n_fft = win = 1024, hop 441, reflect pad 312, 80 Slaney bands 0-8 kHz at 22.05 kHz) is applied.
 * mel_out: device fp32 (B, 80, Tm), Tm = si_mel_frames(N22): the layout si_codebook_splice / si_hifigan_forward take.
 * Needs no weights.  Workspace: si_mel_workspace_bytes. */
int si_mel_frames(int n22);                              /* (n22 + 2*312 - 1024) / 441 + 1, <= 0 when too short */
int si_mel_workspace_bytes(si_ctx* ctx, int B, int N22, size_t* out);
int si_mel_frontend(si_ctx* ctx, const float* wave22, const int32_t* mask_start, const int32_t* mask_end, int normalize,
                    int B, int N22, float* mel_out, void* workspace, size_t workspace_bytes, si_stream_t stream);

/* Ragged batches: clip b holds sample_len[b] (HOST int32 (B)) samples of its row of N22; peak, reflect padding and frame count
 * are the clip's own.  mel_out (B, 80, si_mel_frames(N22)): frames >= si_mel_frames(sample_len[b]) of clip b are zero. */
int si_mel_frontend_varlen(si_ctx* ctx, const float* wave22, const int32_t* mask_start, const int32_t* mask_end, const int32_t* sample_len,
                           int normalize, int B, int N22, float* mel_out, void* workspace, size_t workspace_bytes, si_stream_t stream);

/* Shape helpers (host arithmetic only). */
int si_num_frames(const si_ctx* ctx, int N);            /* encoder frames T for N samples, <0 on error */
int si_vocoder_samples(const si_ctx* ctx, int Tm, int stretch);   /* output samples per clip */

/* Per-kernel timing.  Between si_profile_start and si_profile_stop every kernel launch of this context is
 * bracketed by two HIP events recorded on the launch stream.  si_profile_stop waits for those events and returns
 * one entry per kernel family: launches, summed device milliseconds, summed ALGORITHMIC flops and bytes (layer
 * shapes only: no padding, halo or recompute).  max_launches bounds the event pool; launches beyond it are not
 * recorded.  This is what bench.py's `roofline` object is computed from.  si_profile_filter restricts the bracketing
 * to ONE family (its entry name; NULL or "" = all): an event pair costs ~4 us of stream time, 6 % of a step when every
 * launch carries one, so bench.py times its K steps with only the dominant family bracketed. */
typedef struct si_profile_entry {
    char name[48];
    int32_t launches;
    int32_t reserved;
    double ms;
    double flops;
    double bytes;
} si_profile_entry;
int si_profile_start(si_ctx* ctx, int max_launches);
int si_profile_filter(si_ctx* ctx, const char* family);
int si_profile_stop(si_ctx* ctx, si_profile_entry* out, int capacity, int* count);

/* Test hook.  Intermediates are named "features", "projected", "encoder_in", "last_hidden" (encoder) and
 * "ups<i>", "stage<i>" (vocoder; of the last chunk of clips).  si_debug_capture registers a device buffer that
 * the NEXT forwards copy the named tensor into at the moment it is produced (workspace buffers are recycled
 * within a forward); dst = NULL unregisters.  si_debug_size returns the tensor's float count in the last forward. */
int si_debug_capture(si_ctx* ctx, const char* name, float* dst, long capacity);
long si_debug_size(si_ctx* ctx, const char* name);

#ifdef __cplusplus
}
#endif
#endif /* SI_HIP_H */
