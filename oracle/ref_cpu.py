"""CPU oracle for the I_ea predict hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The shipped path (``speech_inpainting_amd``) never calls into it and fails loudly when the HIP
library is missing.

It is a plain-torch (fp32, CPU) restatement of the reference's arithmetic, one function per row of
SURVEY.md section 8(a).  It imports neither ``transformers`` nor anything from the reference, so it can
travel to the GPU box.  Citations are ``/root/reference/...`` paths unless they start with
``transformers/`` (the third-party package that holds the HuBERT arithmetic; the reference pins
transformers==4.35.0, requirements.txt:8, and calls it at I_ea/model.py:1,32,39-40,82-85).

Pinning: the reference holds no tests or golden vectors for this path, so the oracle
is pinned against outputs of the reference's own modules run in the authoring container
(tools/make_goldens.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # I_ea/hifi_gan/models.py:9


# ----------------------------------------------------------------------------- f-1 (mel front-end)
# Constants of I_ea/dataset/mel_dump.py:11-20.
N_FFT, NUM_MELS, HOP, WIN, MEL_PAD, SR22, FMIN, FMAX = 1024, 80, 441, 1024, 312, 22050, 0.0, 8000.0


def mel_filterbank(sr: int = SR22, n_fft: int = N_FFT, n_mels: int = NUM_MELS, fmin: float = FMIN, fmax: float = FMAX):
    """`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` as called at I_ea/dataset/mel_dump.py:66 (htk=False,
    norm='slaney'): triangular filters on the Slaney mel scale, each scaled by 2 / (f[i+2] - f[i]).
    librosa is absent from this image, so THIS function cannot be pinned against the reference's own dependency
    ("filterbank parity unpinned" in that strict sense); it IS checked against an independent implementation of the same
    filterbank -- `transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")`, the one HuggingFace's
    feature extractors use in librosa's place -- to 1e-9 (tests/test_oracle_golden.py), and by the known-answer
    properties there; everything else in `mel_spectrogram` below is the reference's own torch calls.
    Returns float32 (n_mels, 1 + n_fft // 2)."""
    import numpy as np

    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(y: torch.Tensor) -> torch.Tensor:
    """`get_mel` = `mel_spectrogram` of I_ea/dataset/mel_dump.py:40-98: y (B, n) in [-1, 1] at 22.05 kHz -> (B, 80, Tm).
    Same torch calls in the same order (:71-91); `return_complex=True` + view_as_real is the same STFT arithmetic as
    the script's deprecated `return_complex=False`."""
    basis = torch.from_numpy(mel_filterbank()).to(y.device)
    window = torch.hann_window(WIN, device=y.device)                                         # :68
    y = F.pad(y.unsqueeze(1), (MEL_PAD, MEL_PAD), mode="reflect").squeeze(1)                # :72-73
    spec = torch.view_as_real(torch.stft(y, N_FFT, hop_length=HOP, win_length=WIN, window=window, center=False,
                                         pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)                                            # :89
    spec = torch.matmul(basis, spec)                                                         # :90
    return torch.log(torch.clamp(spec, min=1e-5))                                            # :31,91


def peak_normalize_095(x):
    """`librosa.util.normalize(x) * 0.95` (I_ea/predict.py:93,104) on a float32 numpy clip: divide by max |x| (norm=inf;
    a peak below the dtype's tiny leaves the clip unscaled, librosa's `fill=None` rule), then scale, both in float32."""
    import numpy as np

    x = np.asarray(x, dtype=np.float32)
    m = np.max(np.abs(x)) if x.size else np.float32(0)
    if m < np.finfo(np.float32).tiny:
        m = np.float32(1.0)
    return (x / np.float32(m)) * np.float32(0.95)


def masked_mel(wave22, mask_start: Optional[Sequence[int]], mask_end: Optional[Sequence[int]], normalize: bool = True) -> torch.Tensor:
    """I_ea/predict.py:99-106 for a batch: copy, zero [start, end) (22.05 kHz samples), normalise * 0.95, get_mel."""
    import numpy as np

    clips = []
    for b, w in enumerate(wave22):
        m = np.array(w, dtype=np.float32, copy=True)
        if mask_start is not None:
            m[int(mask_start[b]):int(mask_end[b])] = 0                                       # :102
        clips.append(peak_normalize_095(m) if normalize else m)                              # :104
    return mel_spectrogram(torch.from_numpy(np.stack(clips)))


# ----------------------------------------------------------------------------- f-3 (sample-rate conversion in front of the path)
KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def kaiser_best_window():
    """The `kaiser_best` interpolation filter of resampy -- the resampler behind `librosa.load(path, sr=16000)` at
    I_ea/predict.py:79-80 (librosa==0.9.1, requirements.txt:3: `resample(..., res_type='kaiser_best')` -> `resampy.resample`).
    resampy is NOT in this image and the reference pins no version of it; its published design, restated: the right wing of a
    Kaiser-windowed sinc, `rolloff * sinc(rolloff * t)` on t in [0, num_zeros] sampled 2^precision times per zero crossing,
    tapered by the right half of `kaiser(2 n + 1, beta)` (resampy.filters.sinc_window with the kaiser_best parameters: 64 zero
    crossings, precision 9, roll-off 0.9475937167399596, beta 14.769656459379492).  -> (float64 window (32769,), 512)."""
    import numpy as np
    from scipy.signal.windows import kaiser
    nb = 2 ** KAISER_BEST["precision"]
    n = nb * KAISER_BEST["num_zeros"]
    r = KAISER_BEST["rolloff"]
    sinc_win = r * np.sinc(r * np.linspace(0, KAISER_BEST["num_zeros"], num=n + 1, endpoint=True))
    return kaiser(2 * n + 1, KAISER_BEST["beta"])[n:] * sinc_win, nb


def resample_time_registers(n_out: int, sr_in: int, sr_out: int):
    """resampy's `time_register` for output samples 0 .. n_out - 1: built by REPEATED ADDITION of 1 / ratio in float64, exactly as
    the loop does.  Where the exact time is an integer (every 320th output sample for 22.05 -> 16 kHz) the accumulated rounding
    decides between (n, frac = scale (1 - eps)) and (n + 1, frac = 0), and because the table step is truncated to an integer
    (371 for 371.52) the two give DIFFERENT samples (up to 8 LSB of int16 on speech): bit-faithfulness needs this very sequence."""
    import numpy as np
    tr = np.cumsum(np.full(max(int(n_out), 1), 1.0 / (float(sr_out) / float(sr_in)))) - 1.0 / (float(sr_out) / float(sr_in))
    tr[0] = 0.0
    return tr[:max(int(n_out), 0)]


def resample_kaiser_best(x, sr_in: int, sr_out: int, fix_length: bool = True, first_output: int = 0, first_input: int = 0):
    """`librosa.resample(x, orig_sr, target_sr, res_type='kaiser_best')` of librosa 0.9.1 for a 1-D float clip -- what
    `librosa.load(path, sr=target)` applies to the file's samples (I_ea/predict.py:79-80; also I_ea/metrics.py:82,107).
    resampy's band-limited interpolation loop (resampy/interpn.py `resample_f`), restated per output sample t:
        time = t / ratio (accumulated by repeated addition in the original);  n = int(time);  frac = scale * (time - n)
        left wing : sum_i (win[off + i step] + eta * dwin[off + i step]) * x[n - i],       off, eta = split(frac * 512)
        right wing: sum_k (win[off' + k step] + eta' * dwin[...]) * x[n + 1 + k],          off', eta' = split((scale - frac) * 512)
    with scale = min(1, ratio), step = int(scale * 512), win scaled by ratio when down-sampling, dwin = forward differences;
    n_out = int(n_in * ratio), then librosa pads with zeros to ceil(n_in * ratio) (`util.fix_length`).
    first_output / first_input: `x` is the excerpt x_full[first_input:] of a longer clip and the outputs wanted are samples
    first_output, first_output + 1, ... of the WHOLE clip's conversion (same time registers; taps before the excerpt are missing).
    PINNED by a fixture the reference itself holds: I_ea/hifi_gan/test_files/LJ001-0001_{22k,16k}.wav are one utterance at both
    rates, and floor(32768 * this function(22k file)) equals the 16k file on 154 479 of its 154 480 samples, the other within one
    LSB (tests/test_oracle_golden.py, tests/golden/lj001_resample.npz).  float64 numpy in and out."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    ratio = float(sr_out) / float(sr_in)
    n_in = x.shape[0]
    n_out = int((n_in + first_input) * ratio) - first_output
    win, num_table = kaiser_best_window()
    win = win.copy()
    if ratio < 1:
        win *= ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    step = int(scale * num_table)
    nwin = win.shape[0]
    tr = resample_time_registers(n_out + first_output, sr_in, sr_out)[first_output:] - first_input   # (an integer shift: exact)
    y = np.zeros(n_out)
    taps = np.arange(nwin // step + 2)[None, :]
    for t0 in range(0, n_out, 8192):
        t = np.arange(t0, min(n_out, t0 + 8192))
        n = np.floor(tr[t]).astype(np.int64)
        frac = scale * (tr[t] - n)
        for wing in (0, 1):
            f = frac if wing == 0 else scale - frac
            idx = f * num_table
            off = idx.astype(np.int64)
            eta = idx - off
            cnt = np.minimum(n + 1 if wing == 0 else n_in - n - 1, (nwin - off) // step)[:, None]
            ok = taps < cnt
            wi = np.where(ok, off[:, None] + taps * step, 0)
            xi = np.where(ok, n[:, None] - taps if wing == 0 else n[:, None] + taps + 1, 0)
            y[t] += (((win[wi] + eta[:, None] * delta[wi]) * ok) * x[xi]).sum(1)
    if fix_length and first_output == 0 and first_input == 0:
        full = int(math.ceil(n_in * ratio))
        y = np.concatenate([y, np.zeros(max(full - n_out, 0))])[:full]
    return y


# ----------------------------------------------------------------------------- A0
def mask_and_normalize(wave: torch.Tensor, mask_start: Sequence[int], mask_len: Sequence[int]) -> torch.Tensor:
    """A0.  Zero samples [start, start+len) of each clip (I_ea/predict.py:132-133), then the HF processor's
    zero-mean/unit-variance normalisation ``(x - mean) / sqrt(var + 1e-7)`` with the population variance
    (transformers/models/wav2vec2/feature_extraction_wav2vec2.py:60,66).  wave: (B, N) fp32."""
    x = wave.clone().float()
    for b in range(x.shape[0]):
        s, l = int(mask_start[b]), int(mask_len[b])
        if l > 0:
            x[b, s:s + l] = 0.0
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + 1e-7)


def normalize_padded(waves: Sequence, padding_value: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """The HF processor on clips of DIFFERENT lengths (`padding=True`): each clip is normalised over its own samples, then
    right-padded with `padding_value` (transformers/models/wav2vec2/feature_extraction_wav2vec2.py:78-97: normed =
    (v - v[:len].mean()) / sqrt(v[:len].var() + 1e-7); normed[len:] = padding_value).  -> (input_values (B, Nmax) fp32,
    attention_mask (B, Nmax) int32)."""
    n = max(len(w) for w in waves)
    x = torch.full((len(waves), n), float(padding_value), dtype=torch.float32)
    m = torch.zeros(len(waves), n, dtype=torch.int32)
    for b, w in enumerate(waves):
        v = torch.as_tensor(w, dtype=torch.float32)
        x[b, :len(v)] = (v - v.mean()) / torch.sqrt(v.var(unbiased=False) + 1e-7)
        m[b, :len(v)] = 1
    return x, m


def feature_frame_lengths(arch, sample_lengths: torch.Tensor) -> torch.Tensor:
    """`_get_feat_extract_output_lengths` (modeling_hubert.py:664-677): floor((n - k) / s) + 1 through the conv stack."""
    n = sample_lengths.to(torch.long)
    for k, s in zip(arch.conv_kernel, arch.conv_stride):
        n = torch.div(n - k, s, rounding_mode="floor") + 1
    return n


def mask_samples_from_frames(frame_pos: int, frame_len: int) -> Tuple[int, int]:
    """Sample span the reference zeroes for a frame-level mask:
    ``[pos*320+80, (pos+len)*320+79-80)`` (I_ea/predict.py:133).  Returns (start, length)."""
    s = frame_pos * 320 + 80
    e = (frame_pos + frame_len) * 320 + 79 - 80
    return s, max(e - s, 0)


# ----------------------------------------------------------------------------- weight helpers
def fold_weight_norm(g: torch.Tensor, v: torch.Tensor, dim: int) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim except `dim` (torch.nn.utils.weight_norm)."""
    dims = [d for d in range(v.dim()) if d != dim]
    return v * (g / v.pow(2).sum(dim=dims, keepdim=True).sqrt())


def _get(sd: Dict[str, torch.Tensor], *names: str) -> torch.Tensor:
    for n in names:
        if n in sd:
            return sd[n].float()
    raise KeyError(names[0])


def _conv_weight(sd, prefix: str, dim: int = 0) -> torch.Tensor:
    """Folded or un-folded conv weight (I_ea/hifi_gan/models.py:125-132 remove_weight_norm)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].float()
    if prefix + ".weight_g" in sd:
        return fold_weight_norm(sd[prefix + ".weight_g"].float(), sd[prefix + ".weight_v"].float(), dim)
    g = sd[prefix + ".parametrizations.weight.original0"].float()
    v = sd[prefix + ".parametrizations.weight.original1"].float()
    return fold_weight_norm(g, v, dim)


# ----------------------------------------------------------------------------- A1..A9 HuBERT
def hubert_feature_extractor(sd, arch, x: torch.Tensor, prefix: str = "base_model.") -> torch.Tensor:
    """A1/A1'/A2.  (B, N) normalised wave -> (B, C, T).
    group: conv0 -> GroupNorm(C groups) -> GELU, then conv -> GELU (transformers/models/hubert/modeling_hubert.py:154-175,106-124)
    layer: every layer conv(+bias) -> LayerNorm over channels -> GELU (:127-151)."""
    h = x[:, None, :]
    for i, (k, s) in enumerate(zip(arch.conv_kernel, arch.conv_stride)):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        w = sd[p + "conv.weight"].float()
        b = sd[p + "conv.bias"].float() if arch.conv_bias else None
        h = F.conv1d(h, w, b, stride=s)
        if arch.feat_extract_norm == "group" and i == 0:
            h = F.group_norm(h, w.shape[0], sd[p + "layer_norm.weight"].float(), sd[p + "layer_norm.bias"].float(), 1e-5)
        elif arch.feat_extract_norm == "layer":
            h = F.layer_norm(h.transpose(1, 2), (w.shape[0],), sd[p + "layer_norm.weight"].float(),
                             sd[p + "layer_norm.bias"].float(), 1e-5).transpose(1, 2)
        h = F.gelu(h)
    return h


def hubert_attention(sd, arch, p: str, h: torch.Tensor, key_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A6.  Eager attention: softmax(q k^T * d^-0.5 + mask) v, then out_proj (modeling_hubert.py:234-259,262-344).
    key_mask (B, T) bool: False = padded key, excluded for EVERY query (the additive -inf mask of :250-251, built from the
    frame-level attention mask at :433-437)."""
    B, T, H = h.shape
    nh, hd = arch.num_attention_heads, arch.head_dim
    q = F.linear(h, sd[p + "q_proj.weight"].float(), sd[p + "q_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
    k = F.linear(h, sd[p + "k_proj.weight"].float(), sd[p + "k_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
    v = F.linear(h, sd[p + "v_proj.weight"].float(), sd[p + "v_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
    w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    if key_mask is not None:
        w = w.masked_fill(~key_mask[:, None, None, :], torch.finfo(w.dtype).min)
    w = F.softmax(w, dim=-1)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, T, H)
    return F.linear(o, sd[p + "out_proj.weight"].float(), sd[p + "out_proj.bias"].float())


def hubert_ffn(sd, p: str, h: torch.Tensor) -> torch.Tensor:
    """A7.  Linear -> GELU(erf) -> Linear (modeling_hubert.py:347-368)."""
    h = F.gelu(F.linear(h, sd[p + "intermediate_dense.weight"].float(), sd[p + "intermediate_dense.bias"].float()))
    return F.linear(h, sd[p + "output_dense.weight"].float(), sd[p + "output_dense.bias"].float())


def hubert_pos_conv(sd, arch, prefix: str, h: torch.Tensor) -> torch.Tensor:
    """A4.  Weight-normed (dim=2) grouped Conv1d(k, pad=k//2), drop the last frame for even k, GELU
    (modeling_hubert.py:45-92,95-103)."""
    p = prefix + "encoder.pos_conv_embed.conv"
    w = _conv_weight(sd, p, dim=2)
    k = arch.num_conv_pos_embeddings
    y = F.conv1d(h.transpose(1, 2), w, sd[p + ".bias"].float(), padding=k // 2,
                 groups=arch.num_conv_pos_embedding_groups)
    if k % 2 == 0:
        y = y[:, :, :-1]
    return F.gelu(y).transpose(1, 2)


def hubert_encode(sd, arch, x_norm: torch.Tensor, prefix: str = "base_model.", taps: Optional[dict] = None,
                  attention_mask: Optional[torch.Tensor] = None, output_layer: Optional[int] = None) -> torch.Tensor:
    """A1..A8.  HubertModel.forward in eval mode (modeling_hubert.py:878-947; encoder :407-476 post-LN, :550-623 pre-LN
    'stable').  attention_mask (B, N) 0/1 over SAMPLES (right-padded batches), or None = all ones: the feature extractor
    and the projection see the whole padded input; the frame-level mask (:679-689) then zeroes the padded frames of the
    projected states (:428-431 / :573-576) and excludes them as attention keys."""
    eps = arch.layer_norm_eps
    f = hubert_feature_extractor(sd, arch, x_norm, prefix).transpose(1, 2)          # (B, T, C)
    if taps is not None:
        taps["features"] = f
    p = prefix + "feature_projection."
    if arch.feat_proj_layer_norm:
        f = F.layer_norm(f, (f.shape[-1],), sd[p + "layer_norm.weight"].float(), sd[p + "layer_norm.bias"].float(), eps)
    h = F.linear(f, sd[p + "projection.weight"].float(), sd[p + "projection.bias"].float())
    if taps is not None:
        taps["projected"] = h
    key_mask = None
    if attention_mask is not None:
        flen = feature_frame_lengths(arch, attention_mask.sum(-1))
        key_mask = torch.arange(h.shape[1])[None, :] < flen[:, None]
        h = h * key_mask[:, :, None].to(h.dtype)
    h = h + hubert_pos_conv(sd, arch, prefix, h)
    e = prefix + "encoder."
    H = arch.hidden_size
    if not arch.do_stable_layer_norm:
        h = F.layer_norm(h, (H,), sd[e + "layer_norm.weight"].float(), sd[e + "layer_norm.bias"].float(), eps)
    if taps is not None:
        taps["encoder_in"] = h
    for l in range(arch.num_hidden_layers):
        L = f"{e}layers.{l}."
        if arch.do_stable_layer_norm:
            a = F.layer_norm(h, (H,), sd[L + "layer_norm.weight"].float(), sd[L + "layer_norm.bias"].float(), eps)
            h = h + hubert_attention(sd, arch, L + "attention.", a, key_mask)
            f2 = F.layer_norm(h, (H,), sd[L + "final_layer_norm.weight"].float(), sd[L + "final_layer_norm.bias"].float(), eps)
            h = h + hubert_ffn(sd, L + "feed_forward.", f2)
        else:
            h = h + hubert_attention(sd, arch, L + "attention.", h, key_mask)
            h = F.layer_norm(h, (H,), sd[L + "layer_norm.weight"].float(), sd[L + "layer_norm.bias"].float(), eps)
            h = h + hubert_ffn(sd, L + "feed_forward.", h)
            h = F.layer_norm(h, (H,), sd[L + "final_layer_norm.weight"].float(), sd[L + "final_layer_norm.bias"].float(), eps)
        if taps is not None and l == 0:
            taps["layer0"] = h
        if output_layer is not None and l + 1 == output_layer:
            return h                       # fairseq extract_features(output_layer=L): layer L-1's output, no final LayerNorm
    if arch.do_stable_layer_norm:
        h = F.layer_norm(h, (H,), sd[e + "layer_norm.weight"].float(), sd[e + "layer_norm.bias"].float(), eps)
    return h


# ----------------------------------------------------------------------------- f-2: I_da's encoder call and unit splice
def ida_corrupt(y, frame_start: int, mask_size: int):
    """`y_inpainting = (y + 1e-6) * mask` with mask = 0 on [frame_start, frame_start + mask_size) (I_da/scripts/
    inpainting.py:186-192).  `y` is what `sf.read` returns -- float64 -- so the sum is formed in float64; the float32 cast
    happens in `get_feats` (`torch.from_numpy(x).float()`, I_da/src/hubert_feature_reader.py:50).  -> float64 numpy."""
    import numpy as np

    y = np.asarray(y, dtype=np.float64)
    mask = np.ones_like(y)
    mask[frame_start: frame_start + mask_size] = 0
    return (y + 1e-6) * mask


def hubert_get_feats(sd, arch, signal, output_layer: int, normalize: bool = True, prefix: str = "base_model.") -> torch.Tensor:
    """`HubertFeatureReader.get_feats(None, signal=...)` (I_da/src/hubert_feature_reader.py:44-67) for ONE clip (numpy,
    any float dtype): `.float()` (:50), `F.layer_norm(x, x.shape)` when task.cfg.normalize (:53-54; eps 1e-5, statistics over
    the whole clip), `model.extract_features(source, padding_mask=None, mask=False, output_layer=L)` (:60-65) -> (T, H).
    The clip is shorter than `max_chunk` (1.6 M samples, :13,58), so there is one chunk.
    `extract_features` is fairseq's (`fairseq.models.hubert.HubertModel.extract_features` -> `forward(features_only=True)`
    -> `TransformerEncoder.extract_features(x, padding_mask, tgt_layer=L-1)`): NOT in this image and no version is pinned by
    the reference (requirements.txt omits fairseq).  Its published algorithm, restated: conv feature extractor -> transpose
    -> LayerNorm(512) -> post_extract_proj -> x + GELU(pos_conv(x)) -> [LayerNorm unless layer_norm_first] -> layers
    0..L-1, returning layer L-1's output; the encoder's final LayerNorm (layer_norm_first checkpoints) is applied only when
    NO layer is requested.  (Newer fairseq pads T to a multiple of 2 with masked frames before the layers and strips them
    afterwards: no effect on the real frames.)  That is `hubert_encode(..., output_layer=L)` above -- the transformers port of
    the same architecture, whose `hidden_states[L]` pins it (tests/golden/hidden_layers.npz)."""
    x = torch.as_tensor(signal).float()
    if normalize:
        x = F.layer_norm(x, x.shape)
    return hubert_encode(sd, arch, x.view(1, -1), prefix, None, None, output_layer)[0]


def code_splice(code: torch.Tensor, code_inpainting: torch.Tensor, frame_start: int, mask_size: int, code_hop_size: int = 320) -> torch.Tensor:
    """I_da/scripts/inpainting.py:209-214 on 1-D unit series: the corrupted clip's units survive only inside the mask."""
    out = code_inpainting.clone()
    out[: frame_start // code_hop_size] = code[: frame_start // code_hop_size]
    out[(frame_start + mask_size) // code_hop_size:] = code[(frame_start + mask_size) // code_hop_size:]
    return out


def ida_match_lengths(n_audio: int, n_code: int, n_f0: int, code_hop: int = 320, f0_hop: int = 80):
    """The length bookkeeping of `inpainting()` (I_da/scripts/inpainting.py:219-255): `match_length([(audio, 1), (audio_mask, 1),
    (code, code_hop), (fo, f0_hop)])` (I_da/src/multiseries.py:5-73: units of lcm(hops) samples, the minimum count over the
    series) followed by the removal of `audio % (16 * 80)` samples' worth from the tail of every series.  NOTE that the
    script matches `code` but NOT `code_inpainting` (:219-227), which keeps its full length until the tail removal (:253).
    -> (audio samples, code frames, code_inpainting frames, f0 frames)."""
    import math

    unit = math.lcm(1, 1, code_hop, f0_hop)
    n_unit = min(n_audio // unit, n_code // (unit // code_hop), n_f0 // (unit // f0_hop))
    a, c, ci, f = n_unit * unit, n_unit * (unit // code_hop), n_code, n_unit * (unit // f0_hop)
    to_remove = a % (16 * 80)
    assert to_remove % code_hop == 0                                                         # :245
    if to_remove:
        a, c, ci, f = a - to_remove, c - to_remove // code_hop, ci - to_remove // code_hop, f - to_remove // 80
    return a, c, ci, f


def ida_inpaint(hubert_sd, harch, gen_sd, varch, centroids, emb_c, emb_p, f0_state, wave, frame_start: int, mask_size: int, f0,
                spk_emb=None, output_layer: int = 6, normalize: bool = True, code_hop: int = 320):
    """`inpainting()` of I_da/scripts/inpainting.py:151-266 for ONE clip, from `audio_gt` to the two generator outputs:
    corruption (:186-192), HuBERT features of both signals (:195-198), k-means units (:204-205), unit splice (:209-214),
    length matching (:219-255), `generate` x 2 (:258-259; the CodeGenerator's front + F0 VQ-VAE + unit HiFi-GAN).
    wave: float64 / float32 numpy (N,); f0: (1, Tf0) normalised F0 track (YAAPT + normalize_nonzero, :216-218, is third-party CPU
    code outside the path and enters as an input); spk_emb (E,) or None.  Returns dict(code, code_inpainting, audio_gen,
    audio_inp) with float32 waveforms (before `generate`'s int16 cast)."""
    import numpy as np

    y = np.asarray(wave)
    y_inp = ida_corrupt(y, frame_start, mask_size)
    with torch.no_grad():
        feats = hubert_get_feats(hubert_sd, harch, y, output_layer, normalize)
        feats_inp = hubert_get_feats(hubert_sd, harch, y_inp, output_layer, normalize)
        code = kmeans_assign(feats, centroids)
        code_inp = code_splice(code, kmeans_assign(feats_inp, centroids), frame_start, mask_size, code_hop)
        f0t = torch.as_tensor(f0, dtype=torch.float32).reshape(1, 1, -1)
        _, nc, nci, nf = ida_match_lengths(len(y), code.numel(), f0t.shape[-1], code_hop)
        f0t = f0t[..., :nf]
        z_p = f0_vq_codes(f0_encoder_forward(f0_state, f0t), f0_state["vq.level_blocks.0.k"])
        spk = None if spk_emb is None else torch.as_tensor(spk_emb, dtype=torch.float32).reshape(1, -1)
        outs = []
        for c in (code[:nc], code_inp[:nci]):
            x = code_generator_front(c[None], emb_c, z_p, emb_p, spk)
            outs.append(generator_forward(gen_sd, varch, x)[0, 0])
    return {"code": code[:nc], "code_inpainting": code_inp[:nci], "feats": feats, "feats_inpainting": feats_inp,
            "audio_gen": outs[0], "audio_inp": outs[1]}


def custom_model_forward(sd, arch, x_norm: torch.Tensor, taps: Optional[dict] = None,
                         attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A9.  CustomModel.forward(input_values, attention_mask): final_layers = LayerNorm(H) -> Linear(H, codebook_dim) on
    last_hidden_state (I_ea/model.py:75-78,80-89).  Returns (B, T, codebook_dim)."""
    h = hubert_encode(sd, arch, x_norm, "base_model.", taps, attention_mask)
    if taps is not None:
        taps["last_hidden"] = h
    h = F.layer_norm(h, (arch.hidden_size,), sd["final_layers.0.weight"].float(), sd["final_layers.0.bias"].float(), 1e-5)
    return F.linear(h, sd["final_layers.1.weight"].float(), sd["final_layers.1.bias"].float())


# ----------------------------------------------------------------------------- A10..A13 codebook
def gather_masked_frames(outputs: torch.Tensor, frame_pos: Sequence[int], lm: int) -> torch.Tensor:
    """A10.  values[i] = outputs[i, pos_i : pos_i + Lm] (I_ea/predict.py:164-168)."""
    return torch.stack([outputs[i, int(frame_pos[i]):int(frame_pos[i]) + lm] for i in range(outputs.shape[0])])


def codebook_tables(centroids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """A11.  centroids (K, D) = cluster_centers_.  Returns (center_ (D,), centred (K, D))
    (I_ea/loss_fn.py:10-14; C = cluster_centers_.T at I_ea/dataset/km_label.py:14)."""
    c = centroids.float()
    center = c.mean(dim=0)
    return center, c - center[None, :]


def codebook_argmax(values: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    """A12.  pred = argmax_k cos(v, C_k - mean(C)); v is NOT centred (I_ea/loss_fn.py:44-47).
    values (B, Lm, D) -> (B, Lm) int64."""
    _, cc = codebook_tables(centroids)
    flat = values.reshape(-1, values.shape[-1]).float()
    sim = F.cosine_similarity(flat.unsqueeze(1), cc.unsqueeze(0), dim=-1)
    return torch.argmax(sim, dim=1).view(values.shape[:-1])


def kmeans_assign(feats: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    """f-2.  `kmeans_model.predict(feats)` (I_da/scripts/inpainting.py:204-205; sklearn minimises ||c||^2 - 2 x.c, the
    ||x||^2 term being common) == `ApplyKmeans.__call__` (I_ea/dataset/km_label.py:20-24: x^2 - 2 x C + C^2, argmin).
    feats (rows, D), centroids (K, D) -> int64 (rows)."""
    x, c = feats.float(), centroids.float()
    dist = x.pow(2).sum(1, keepdim=True) - 2 * torch.matmul(x, c.T) + c.pow(2).sum(1)[None, :]
    return dist.argmin(dim=1)


def mel_signal_metrics(t1: torch.Tensor, t2: torch.Tensor, center: torch.Tensor):
    """f-4.  `Metrics.avg_cosine_sim`, `.avg_d2_dist`, `.rmse` (I_ea/metrics.py:38-62) on mel segments (80, L), same torch
    calls in the same order; center = the (80,) vector the constructor stores as `centroids.unsqueeze(1)` (:26).
    PARITY UNPINNED: I_ea/metrics.py cannot be imported (librosa / pystoi / pesq / speech_recognition absent, and its
    `from torch.nn.functional import F` line fails on its own); these are its statements restated."""
    c = center.reshape(-1, 1)
    cos = F.cosine_similarity(t1 - c, t2 - c, dim=0).mean()                                   # :39-42
    log_scale = 20 / torch.log(torch.tensor(10.0))                                            # :45
    a, b = t1 - torch.mean(t1, dim=0), t2 - torch.mean(t2, dim=0)                             # :46-49
    d2 = (log_scale * torch.sqrt(torch.mean((a - b) ** 2, dim=0))).mean()                     # :50-53
    r = log_scale * torch.sqrt(torch.mean((a - b) ** 2))                                      # :61
    return cos, d2, r


def sisdr(x_est, x_ref) -> float:
    """f-4.  `Metrics.sisdr` (I_ea/metrics.py:127-142), numpy, in the dtype of the inputs."""
    import numpy as np

    eps = np.finfo(x_est.dtype).eps
    reference = x_ref.reshape(x_ref.size, 1)
    estimate = x_est.reshape(x_est.size, 1)
    rss = np.dot(reference.T, reference)
    a = (eps + np.dot(reference.T, estimate)) / (rss + eps)
    e_true = a * reference
    e_res = estimate - e_true
    return float(10 * np.log10((eps + (e_true ** 2).sum()) / (eps + (e_res ** 2).sum())))


def code_generator_front(code: torch.Tensor, emb_c: torch.Tensor, f0_code: Optional[torch.Tensor] = None,
                         emb_p: Optional[torch.Tensor] = None, spk_emb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f-2.  The front of `CodeGenerator.forward` (I_da/src/model.py:148-189, LUT configuration): embedding look-ups
    (:153,161), `_upsample` of the shorter series (:79-119: unsqueeze + repeat(max // len) + view, i.e. frame-wise
    repeat; a length that does not divide raises NotImplementedError at :110-112), channel concat (:169), speaker
    embedding repeated over the frames (:175-176).  PARITY UNPINNED: `src.model` is not importable here (fairseq,
    kaldi_io absent) and the reference holds no fixture for it; restated from the source text only."""
    def upsample(sig: torch.Tensor, frames: int) -> torch.Tensor:
        if sig.dim() == 2:
            sig = sig.unsqueeze(2)
        b, ch, t = sig.shape
        rep = frames // t
        sig = sig.unsqueeze(3).repeat(1, 1, 1, rep)
        if (frames - t * rep) // rep > 0:
            raise NotImplementedError("Padding condition signal - misalignment between condition features.")
        return sig.reshape(b, ch, frames)

    x = emb_c[code].transpose(1, 2)
    if f0_code is not None:
        p = emb_p[f0_code].transpose(1, 2)
        if x.shape[-1] < p.shape[-1]:
            x = upsample(x, p.shape[-1])
        else:
            p = upsample(p, x.shape[-1])
        x = torch.cat([x, p], dim=1)
    if spk_emb is not None:
        x = torch.cat([x, upsample(spk_emb, x.shape[-1])], dim=1)
    return x


def f0_encoder_forward(state: dict, f0: torch.Tensor, down_t: int = 4, stride_t: int = 2, depth: int = 4,
                       dilation_growth: int = 3) -> torch.Tensor:
    """f-2.  `FoVQVAE.encoder(fo)` as run inside `CodeGenerator.forward` (I_da/src/model.py:160-163): jukebox.py `Encoder`
    :200-262 with one level of `EncoderConvBlock` (:11-116): down_t x [Conv1d(k = 2 s, stride s, pad s / 2) (:54-68, odd s:
    2 s + 1 / s // 2 + 1) -> Resnet1D (resnet.py:57-97): depth x `x + Conv1d_k1(ReLU(Conv1d_k3(ReLU(x), dilation =
    growth ** j, padding = dilation)))` (:29-54, res_scale 1)] -> Conv1d(width, out, 3, 1, 1) (:80-82).
    f0 (B, 1, T) -> (B, out_width, T // stride ** down_t).  Pinned by tests/golden/f0_vqvae.npz: outputs of the reference's
    own `Encoder` loaded by file path (tools/make_goldens.py::f0_vqvae_cases), reproduced bit for bit."""
    pre = "encoder.level_blocks.0.model."
    k, pad = (2 * stride_t, stride_t // 2) if stride_t % 2 == 0 else (2 * stride_t + 1, stride_t // 2 + 1)
    x = f0.float()
    for i in range(down_t):
        x = F.conv1d(x, state[f"{pre}{i}.0.weight"].float(), state[f"{pre}{i}.0.bias"].float(), stride=stride_t, padding=pad)
        assert state[f"{pre}{i}.0.weight"].shape[-1] == k
        for j in range(depth):
            r = f"{pre}{i}.1.model.{j}.model."
            dil = dilation_growth ** j
            y = F.conv1d(F.relu(x), state[r + "1.weight"].float(), state[r + "1.bias"].float(), padding=dil, dilation=dil)
            x = x + F.conv1d(F.relu(y), state[r + "3.weight"].float(), state[r + "3.bias"].float())
    return F.conv1d(x, state[f"{pre}{down_t}.weight"].float(), state[f"{pre}{down_t}.bias"].float(), padding=1)


def f0_vq_codes(h: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """f-2.  `BottleneckBlock.encode` (I_da/src/modules/vq.py:133-144): h (N, C, T) -> permute / flatten (:92-95) ->
    arg-min over bins of |x|^2 - 2 x.k^T + |k|^2 (:117-127) -> (N, T) int64.  Pinned by tests/golden/f0_vqvae.npz (the
    reference's `Bottleneck.forward` in eval mode, which is what `CodeGenerator.forward` calls at model.py:164)."""
    n, c, t = h.shape
    x = h.permute(0, 2, 1).reshape(-1, c).float()
    kw = k.float().t()
    dist = (x ** 2).sum(-1, keepdim=True) - 2 * x @ kw + (kw ** 2).sum(0, keepdim=True)
    return dist.argmin(-1).reshape(n, t)


def cos_sim_loss(values: torch.Tensor, labels: torch.Tensor, centroids: torch.Tensor):
    """f-4.  `LossFunction.cos_sim` in full (I_ea/loss_fn.py:29-47): loss = -sum(cos(v, centred target) - 1), pred =
    arg-max labels; plus `cos_sim_target_labels` (:49-62).  values (B, Lm, D), labels (B, Lm) ->
    (loss scalar, pred (B, Lm), cos_pred_target (B*Lm,))."""
    _, cc = codebook_tables(centroids)
    flat = values.reshape(-1, values.shape[-1]).float()
    tgt = cc[labels.reshape(-1)]
    loss = -(F.cosine_similarity(flat, tgt) - 1).sum()
    pred = codebook_argmax(values, centroids)
    cpt = F.cosine_similarity(cc[pred.reshape(-1)].T, cc[labels.reshape(-1)].T, dim=0)
    return loss, pred, cpt


def splice_centroids(mel: torch.Tensor, labels: torch.Tensor, centroids: torch.Tensor, frame_pos: Sequence[int]) -> torch.Tensor:
    """A13.  mel[b, :, pos:pos+Lm] = (centred[labels] + center_).T (I_ea/predict.py:184-187).  mel (B, D, Tm)."""
    center, cc = codebook_tables(centroids)
    out = mel.clone().float()
    lm = labels.shape[1]
    for b in range(mel.shape[0]):
        p = int(frame_pos[b])
        out[b, :, p:p + lm] = (cc[labels[b]] + center).T
    return out


# ----------------------------------------------------------------------------- A14
def extend_mel(spec: torch.Tensor) -> torch.Tensor:
    """A14.  Time-only bilinear stretch by 441/256, align_corners=False
    (I_ea/hifi_gan/inference_modified.py:16-19).  (B, D, Tm) -> (B, D, floor(Tm*441/256)).
    Restated without F.interpolate: src = fma(dst + 0.5, float(256/441), -0.5) clamped at 0, two-tap lerp."""
    B, D, Tm = spec.shape
    scale = 441 / 256
    Tout = int(math.floor(float(Tm) * scale))
    rscale = torch.tensor(1.0 / scale, dtype=torch.float32)                      # static_cast<float>(1.0 / scale)
    dst = torch.arange(Tout, dtype=torch.float32)
    # ATen evaluates scale * (dst + 0.5) - 0.5 as ONE fused multiply-add in fp32 (verified against
    # F.interpolate bit patterns: a separate multiply loses up to 4e-6 of lambda near index 32+).
    # The double product of two floats is exact, so rounding the double result once reproduces the fma.
    src = ((dst + 0.5).double() * rscale.double() - 0.5).float().clamp(min=0.0)
    i0 = src.floor().to(torch.int64).clamp_max(Tm - 1)
    i1 = torch.clamp(i0 + 1, max=Tm - 1)
    l1 = (src - i0.float()).clamp(0.0, 1.0)
    l0 = 1.0 - l1
    x = spec.float()
    return l0 * x[:, :, i0] + l1 * x[:, :, i1]


# ----------------------------------------------------------------------------- B1..B5 HiFi-GAN
def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """I_ea/hifi_gan/utils.py:47-48."""
    return int((kernel_size * dilation - dilation) / 2)


def generator_forward(sd, arch, mel: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """B1..B4.  Generator.forward with ResBlock1 or ResBlock2 blocks (I_ea/hifi_gan/models.py:107-123,36-43,63-68).
    mel (B, 80, Tm') -> (B, 1, Tm'*hop) in (-1, 1).  Accepts folded or weight-normed state dicts (B5)."""
    x = F.conv1d(mel.float(), _conv_weight(sd, "conv_pre"), sd["conv_pre.bias"].float(), padding=3)
    nk = len(arch.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(arch.upsample_rates, arch.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _conv_weight(sd, f"ups.{i}"), sd[f"ups.{i}.bias"].float(), stride=u, padding=(k - u) // 2)
        if taps is not None:
            taps[f"ups{i}"] = x
        xs = None
        for j, (rk, dil) in enumerate(zip(arch.resblock_kernel_sizes, arch.resblock_dilation_sizes)):
            r = f"resblocks.{i * nk + j}."
            y = x
            if str(getattr(arch, "resblock", "1")) == "2":      # ResBlock2.forward (models.py:63-68): x = x + c(lrelu(x))
                for n, d in enumerate(dil):
                    t = F.conv1d(F.leaky_relu(y, LRELU_SLOPE), _conv_weight(sd, f"{r}convs.{n}"), sd[f"{r}convs.{n}.bias"].float(),
                                 dilation=d, padding=get_padding(rk, d))
                    y = t + y
                xs = y if xs is None else xs + y
                continue
            for n, d in enumerate(dil):
                t = F.leaky_relu(y, LRELU_SLOPE)
                t = F.conv1d(t, _conv_weight(sd, f"{r}convs1.{n}"), sd[f"{r}convs1.{n}.bias"].float(),
                             dilation=d, padding=get_padding(rk, d))
                t = F.leaky_relu(t, LRELU_SLOPE)
                t = F.conv1d(t, _conv_weight(sd, f"{r}convs2.{n}"), sd[f"{r}convs2.{n}.bias"].float(),
                             padding=get_padding(rk, 1))
                y = t + y
            xs = y if xs is None else xs + y
        x = xs / nk
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x)                          # default slope 0.01 (models.py:119)
    x = F.conv1d(x, _conv_weight(sd, "conv_post"), sd["conv_post.bias"].float(), padding=3)
    return torch.tanh(x)


def to_int16_pcm(audio: torch.Tensor):
    """B6.  ``audio * 32768`` then numpy ``astype('int16')`` (truncation toward zero) (I_ea/predict.py:204-206)."""
    return (audio * 32768.0).cpu().numpy().astype("int16")


# ----------------------------------------------------------------------------- whole path
def predict_batch(hubert_sd, harch, gen_sd, varch, centroids, wave16: torch.Tensor, mel: torch.Tensor,
                  frame_pos: Sequence[int], frame_len: int, blind: bool = False, taps: Optional[dict] = None):
    """The I_ea predict path for a batch (I_ea/predict.py:130-207, batch-ified).

    wave16 (B, N) raw 16 kHz clips; mel (B, 80, Tm) log-mel of the (masked) 22.05 kHz clip;
    frame_pos (B,) first masked 20 ms frame; frame_len Lm.  blind=True replaces every encoder frame
    (mask_pos=0, mask_len=T).  Returns dict(feats, labels, mel, wave)."""
    B = wave16.shape[0]
    with torch.no_grad():
        if blind:
            starts, lens = [0] * B, [0] * B
        else:
            sl = [mask_samples_from_frames(int(p), frame_len) for p in frame_pos]
            starts, lens = [s for s, _ in sl], [l for _, l in sl]
        x = mask_and_normalize(wave16, starts, lens)
        feats = custom_model_forward(hubert_sd, harch, x, taps)
        T = feats.shape[1]
        if blind:
            pos, lm = [0] * B, min(T, mel.shape[2])
        else:
            pos, lm = [int(p) for p in frame_pos], frame_len
        values = gather_masked_frames(feats, pos, lm)
        labels = codebook_argmax(values, centroids)
        mel2 = splice_centroids(mel, labels, centroids, pos)
        wav = generator_forward(gen_sd, varch, extend_mel(mel2), taps)
    return {"feats": feats, "labels": labels, "mel": mel2, "wave": wav[:, 0, :]}
