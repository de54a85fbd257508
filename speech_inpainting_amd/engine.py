"""Host-side mirror of the reference's module interface over the HIP library.

The reference's seam is Python duck-typing at three calls; the classes below keep those
names, argument meanings and shapes, and forward to libsi_hip.so:

    model(input_values (B, N), attention_mask)      -> (B, T, 80)        I_ea/model.py:80-89
    generator(feats (B, 80, T'))                     -> (B, 1, T' * 256)  I_ea/hifi_gan/models.py:107-123
    loss.cos_sim(values, labels)[1]                  -> predicted labels  I_ea/loss_fn.py:44-47

plus `InpaintingEngine.predict_batch`, the batched form of the script body I_ea/predict.py:130-207, which is what the
benchmark times.  Nothing here computes on the CPU: a missing library or a CPU device raises.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Sequence

import torch

from .arch import HubertArch, VocoderArch
from .checkpoint import flatten_checkpoint
from .native import NativeContext, make_desc


def mask_samples_from_frames(frame_pos: int, frame_len: int):
    """Sample span zeroed for a frame-level mask: [pos*320+80, (pos+len)*320+79-80) (I_ea/predict.py:133)."""
    s = frame_pos * 320 + 80
    e = (frame_pos + frame_len) * 320 + 79 - 80
    return s, max(e - s, 0)


def ida_match_lengths(n_audio: int, n_code: int, n_f0: int, code_hop: int = 320, f0_hop: int = 80):
    """Length bookkeeping of I_da's `inpainting()` (I_da/scripts/inpainting.py:219-255): `match_length` over (audio, 1),
    (audio_mask, 1), (code, code_hop), (fo, f0_hop) -- whole units of lcm(hops) samples, the minimum count over the series
    (I_da/src/multiseries.py:33-52) -- then `audio % (16 * 80)` samples' worth removed from every tail (:243-255).  The script
    matches `code` but not `code_inpainting`, which is only tail-trimmed.  -> (audio samples, code frames, code_inpainting
    frames, f0 frames)."""
    import math
    unit = math.lcm(code_hop, f0_hop)
    n_unit = min(n_audio // unit, n_code // (unit // code_hop), n_f0 // (unit // f0_hop))
    a, c, ci, f = n_unit * unit, n_unit * (unit // code_hop), n_code, n_unit * (unit // f0_hop)
    to_remove = a % (16 * 80)
    if to_remove % code_hop:
        raise AssertionError(f"to_remove={to_remove} is not a multiple of code_hop_size={code_hop} (I_da/scripts/inpainting.py:245)")
    if to_remove:
        a, c, ci, f = a - to_remove, c - to_remove // code_hop, ci - to_remove // code_hop, f - to_remove // 80
    if min(c, ci, f) <= 0:
        raise ValueError(f"clip too short for I_da's length matching: audio {n_audio}, code {n_code}, f0 {n_f0} frames")
    return a, c, ci, f


class InpaintingEngine:
    """One model pair (HuBERT + head, codebook, HiFi-GAN generator) resident on one GPU."""

    def __init__(self, harch: HubertArch, varch: VocoderArch, num_clusters: int, device="cuda:0",
                 encoder_dtype: str = "fp32", vocoder_dtype: str = "fp32", vocoder_chunk: int = 0):
        self.harch, self.varch = harch, varch
        self.device = torch.device(device)
        self.encoder_dtype, self.vocoder_dtype = encoder_dtype, vocoder_dtype
        self.ctx = NativeContext(make_desc(harch, varch, num_clusters, encoder_dtype, vocoder_dtype, vocoder_chunk), self.device)
        self._resamplers = {}

    # ---- weights
    def load_state(self, hubert_sd: Mapping[str, torch.Tensor], gen_sd: Mapping[str, torch.Tensor], codebook: Optional[torch.Tensor] = None):
        """hubert_sd: a CustomModel state dict, or the encoder alone (a HuggingFace directory / an I_da feature reader has no
        `final_layers`).  codebook (K, codebook_dim) or None (I_da: the unit codebook lives in HuBERT feature space and is passed
        per call).  An engine without a trained head or without a codebook serves the encoder-only entry points
        (`extract_features`, `ida_inpaint_batch`) and the vocoder; the I_ea calls that need the missing part (`encode`,
        `predict_batch`, `splice`, `codebook_metrics`, ...) raise instead of computing on placeholders -- the reference's predict
        path always loads the CustomModel .pt with its head (I_ea/predict.py:149) and its k-means model (:66-70)."""
        from .checkpoint import fresh_final_layers, normalize_hubert_keys
        hubert_sd = normalize_hubert_keys(hubert_sd)
        self._has_head = "final_layers.1.weight" in hubert_sd
        if not self._has_head:
            hubert_sd.update(fresh_final_layers(self.harch))          # placeholder so that the packed layout is complete; never served
        self._has_codebook = codebook is not None
        if codebook is None:
            codebook = torch.zeros(self.ctx.desc.num_clusters, self.harch.codebook_dim)
        blob, index = flatten_checkpoint(hubert_sd, gen_sd, codebook)
        self.ctx.load_weights(blob, index)
        return self

    def alloc_weights(self, has_head: bool = True, has_codebook: bool = True):
        """Receiving rank of the weight broadcast: the flags say what the SOURCE rank's checkpoint held."""
        self.ctx.alloc_weights()
        self._has_head, self._has_codebook = bool(has_head), bool(has_codebook)
        return self

    def _need(self, head: bool = False, codebook: bool = False, what: str = "this call"):
        if head and not getattr(self, "_has_head", True):
            raise RuntimeError(f"{what} needs the trained `final_layers` head, but the loaded checkpoint held the encoder only "
                               "(load the CustomModel .pt, I_ea/predict.py:149)")
        if codebook and not getattr(self, "_has_codebook", True):
            raise RuntimeError(f"{what} needs the k-means codebook, but none was loaded (load_state(..., codebook=...))")

    def weights_tensor(self) -> torch.Tensor:
        return self.ctx.weights_tensor()

    def weights_check(self) -> None:
        """After the weight broadcast: the blob in this context carries the fingerprint of THIS context's layout (raises otherwise)."""
        self.ctx.weights_check()

    # ---- the three stages
    def encode(self, wave16: torch.Tensor, mask_start: Optional[torch.Tensor] = None, mask_len: Optional[torch.Tensor] = None,
               normalize: bool = True, valid_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """valid_len (B,) int32: real samples per clip of a RIGHT-PADDED batch (the reference's attention_mask.sum(-1))."""
        self._need(head=True, what="encode")
        return self.ctx.hubert_forward(wave16, mask_start, mask_len, normalize, valid_len)

    def extract_features(self, wave16: torch.Tensor, output_layer: int, normalize="layer_norm",
                         mask_start: Optional[torch.Tensor] = None, mask_len: Optional[torch.Tensor] = None,
                         pre_mask_add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`HubertFeatureReader.get_feats` for a batch (I_da/src/hubert_feature_reader.py:44-67): (B, N) raw clips ->
        (B, T, H) hidden state after `output_layer` transformer layers; see NativeContext.hubert_extract_features."""
        return self.ctx.hubert_extract_features(wave16, output_layer, normalize, mask_start, mask_len, pre_mask_add)

    def ida_inpaint_batch(self, wave16: torch.Tensor, frame_start, mask_size: int, centroids: torch.Tensor,
                          generator: "CodeGenerator", f0: torch.Tensor, emb: Optional[torch.Tensor] = None,
                          output_layer: int = 6, normalize: bool = True, code_hop_size: int = 320, f0_hop: int = 80) -> Dict[str, torch.Tensor]:
        """`inpainting()` of I_da/scripts/inpainting.py:151-266 for a batch of equal-length clips, on this GPU end to end:
        corruption `(y + 1e-6) * mask` (:186-192) fused into the encoder's first conv, HuBERT features of the clean and the
        corrupted clips at `output_layer` (:195-198; ONE encoder pass over 2B clips), k-means units (:204-205, GPU instead of
        sklearn on the host), unit splice (:209-214), the script's length bookkeeping (:219-255), and `generate` for both unit
        series (:258-259; one CodeGenerator pass over 2B series).
        wave16 (B, N) fp32 at 16 kHz; frame_start int or (B,) int tensor (samples; the script uses 1.5 s, :188); mask_size
        samples; centroids (K, H) the k-means model's `cluster_centers_`; generator a `CodeGenerator` over this engine (with
        its `F0Quantizer`); f0 (B, 1, Tf0) the normalised F0 track (YAAPT + normalize_nonzero, :216-217, is third-party CPU code
        outside the path); emb (B, E) speaker embedding or None.
        -> dict(code (B, F), code_inpainting (B, F'), audio_gen (B, F * hop), audio_inp (B, F' * hop), feats (2B, T, H))."""
        dev = self.device
        B, N = wave16.shape
        fs = torch.as_tensor(frame_start, dtype=torch.int32, device=dev).reshape(-1).expand(B).contiguous()
        zi = torch.zeros(B, dtype=torch.int32, device=dev)
        ms = torch.cat([zi, fs])
        ml = torch.cat([zi, torch.full((B,), int(mask_size), dtype=torch.int32, device=dev)])
        add = torch.cat([torch.zeros(B, dtype=torch.float64, device=dev), torch.full((B,), 1e-6, dtype=torch.float64, device=dev)])
        if generator.f0_quantizer is None:
            raise ValueError("ida_inpaint_batch: the CodeGenerator needs its F0Quantizer (the fixed F0 VQ-VAE, I_da/src/model.py:160-166)")
        # The F0 VQ-VAE's conv encoder (one workgroup per track: 16 of the chip's 256 CUs for ~0.35 ms) depends on nothing the HuBERT
        # encoder computes: it runs on a side stream UNDER the encoder and is joined before its bottleneck's arg-min.
        _, nc, nci, nf = ida_match_lengths(N, self.ctx.num_frames(N), f0.shape[-1], code_hop_size, f0_hop)
        f0 = f0.to(dev, torch.float32)[..., :nf].contiguous()
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(dev)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            f0_feats = generator.f0_quantizer.features(f0)
        both = torch.cat([wave16, wave16]).contiguous()
        hid = self.extract_features(both, output_layer, "layer_norm" if normalize else None, ms, ml, add)       # (2B, T, H)
        T, H = hid.shape[1], hid.shape[2]
        units = self.ctx.kmeans_assign(hid.reshape(2 * B * T, H), centroids.to(dev, torch.float32).contiguous()).reshape(2 * B, T)
        code = units[:B].contiguous()
        first = torch.div(fs, code_hop_size, rounding_mode="floor").to(torch.int32)
        last = torch.div(fs + int(mask_size), code_hop_size, rounding_mode="floor").to(torch.int32)
        code_inp = self.ctx.code_splice(code, units[B:].contiguous(), first, last)
        main.wait_stream(self._side)
        f0_feats.record_stream(main)
        z_p = generator.f0_quantizer.codes(f0_feats)                                   # the same F0 track conditions both outputs
        code, code_inp = code[:, :nc].contiguous(), code_inp[:, :nci].contiguous()
        if nc == nci:
            wav = generator(code=torch.cat([code, code_inp]), f0_code=torch.cat([z_p, z_p]),
                            emb=None if emb is None else torch.cat([emb, emb]))[:, 0]
            gen, inp = wav[:B], wav[B:]
        else:                                   # the script trims `code` but not `code_inpainting` (:219-227): two shapes
            gen = generator(code=code, f0_code=z_p, emb=emb)[:, 0]
            inp = generator(code=code_inp, f0_code=z_p, emb=emb)[:, 0]
        return {"code": code, "code_inpainting": code_inp, "audio_gen": gen, "audio_inp": inp, "feats": hid}

    def splice(self, feats: torch.Tensor, frame_pos: torch.Tensor, lm: int, mel: torch.Tensor) -> torch.Tensor:
        self._need(codebook=True, what="splice")
        return self.ctx.codebook_splice(feats, frame_pos, lm, mel)

    def splice_labels(self, labels: torch.Tensor, frame_pos: torch.Tensor, mel: torch.Tensor) -> None:
        """`expected_inpaint`'s splice (I_ea/predict.py:177-189): the raw centroids of GIVEN labels (B, Lm) into mel, in place."""
        self._need(codebook=True, what="splice_labels")
        self.ctx.codebook_splice_labels(labels, frame_pos, mel)

    def vocode(self, mel: torch.Tensor, stretch: bool = True) -> torch.Tensor:
        return self.ctx.hifigan_forward(mel, stretch)

    # ---- the script's three generator passes (I_ea/predict.py:123-128,196-207) differ only around the mask
    def receptive_radius(self) -> int:
        """Output samples (one side) an input frame of the generator can reach, from the architecture: conv_pre (k 7) 3 frames,
        per stage the transposed conv (ceil(k / u) + 1 input rows) and the widest ResBlock (sum over its dilations of (k - 1) / 2 *
        (d + 1) rows for ResBlock1, (k - 1) / 2 * d for ResBlock2), conv_post 3 samples; conservative (whole rows)."""
        v = self.varch
        hop = 1
        for u in v.upsample_rates:
            hop *= u
        r, per_row = 3 * hop, hop                                    # conv_pre at the frame rate
        two = str(v.resblock) == "1"
        for u, k in zip(v.upsample_rates, v.upsample_kernel_sizes):
            r += (-(-k // u) + 1) * per_row                           # the transposed conv, in rows of its input
            per_row //= u
            r += max(sum((rk - 1) // 2 * (d + (1 if two else 0)) for d in dil) for rk, dil in zip(v.resblock_kernel_sizes, v.resblock_dilation_sizes)) * per_row
        return r + 3

    def vocode_window(self, wave_base: torch.Tensor, mel_var: torch.Tensor, frame_pos: Sequence[int], lm: int) -> torch.Tensor:
        """The waveform of `mel_var` (B, 80, Tm), which differs from the mel that produced `wave_base` ONLY in frames
        [frame_pos[b], frame_pos[b] + lm): the generator runs on a window of the stretched frames around the change (as a ragged
        batch: a window clamped at a clip edge keeps that edge's real zero padding) and the samples the change can reach are spliced
        into a copy of `wave_base`.  Every output sample of the generator is the same fixed-order sum wherever its tile falls, so
        the result is BIT-IDENTICAL to a full pass (asserted per vocoder mode in tests/test_gpu_configs.py)."""
        import math
        ext = self.ctx.extend_mel(mel_var.contiguous())
        B, D, Tout = ext.shape
        hop = wave_base.shape[1] // Tout
        R = self.receptive_radius()
        Rf = -(-R // hop)
        r = 441.0 / 256.0
        spans = []
        for b in range(B):
            p = int(frame_pos[b])
            c0 = max(int(math.floor((p - 0.5) * r - 0.5)) - 1, 0)      # stretched frames whose two source frames touch [p, p + lm)
            c1 = min(int(math.ceil((p + lm + 0.5) * r - 0.5)) + 1, Tout)
            spans.append((max(c0 - 2 * Rf, 0), min(c1 + 2 * Rf, Tout)))
        W = max(w1 - w0 for w0, w1 in spans)
        win = torch.zeros(B, D, W, dtype=torch.float32, device=self.device)
        for b, (w0, w1) in enumerate(spans):
            win[b, :, :w1 - w0] = ext[b, :, w0:w1]
        out = self.vocode_ragged(win, [w1 - w0 for w0, w1 in spans], stretch=False)
        wave = wave_base.clone()
        for b, (w0, w1) in enumerate(spans):
            s0 = w0 * hop + (Rf * hop if w0 > 0 else 0)               # a window edge that is not a clip edge: its first / last Rf frames
            s1 = w1 * hop - (Rf * hop if w1 < Tout else 0)            # see missing neighbours and are dropped
            wave[b, s0:s1] = out[b, s0 - w0 * hop:s1 - w0 * hop]
        return wave

    def mel(self, wave22: torch.Tensor, mask_start: Optional[torch.Tensor] = None, mask_end: Optional[torch.Tensor] = None,
            normalize: bool = True) -> torch.Tensor:
        """Vocoder-side front-end (I_ea/predict.py:99-106): zero [mask_start, mask_end) of each raw 22.05 kHz clip,
        peak-normalise * 0.95, log-mel -> (B, 80, Tm)."""
        return self.ctx.mel_frontend(wave22, mask_start, mask_end, normalize)

    def resample(self, x: torch.Tensor, sr_in: int, sr_out: int, kind: str = "kaiser_best", lens=None) -> torch.Tensor:
        """(B, n) fp32 clips at sr_in -> (B, ceil(n * sr_out / sr_in)) at sr_out on the GPU: `librosa.load(..., sr=...)`'s resampling
        (I_ea/predict.py:79-80).  kind="kaiser_best" (default): resampy's band-limited interpolation, what librosa 0.9.1 runs --
        pinned bit for bit by the reference-held LJ001-0001 22k / 16k pair (si_resample_sinc).  kind="poly": the polyphase Kaiser
        FIR with scipy.signal.resample_poly's arithmetic (si_resample_poly; a different filter).  lens: per-clip sample counts of a
        ragged batch (kaiser_best only): clip b's output is zero past int(lens[b] * ratio)."""
        from . import audio
        if sr_in == sr_out:
            return x.clone()
        key = (kind, int(sr_in), int(sr_out), int(x.shape[1]))
        if kind == "kaiser_best":
            if key not in self._resamplers:
                f = audio.design_kaiser_best(sr_in, sr_out, x.shape[1])
                for k in ("win", "dwin", "time_reg"):
                    f[k] = torch.from_numpy(f[k]).to(self.device)
                self._resamplers[key] = f
            f = self._resamplers[key]
            n_len = None if lens is None else torch.as_tensor(list(lens), dtype=torch.int32).to(self.device)
            return self.ctx.resample_sinc(x.contiguous(), f, f["n_out"], n_len)
        if kind != "poly":
            raise ValueError(f"resample kind {kind!r}: 'kaiser_best' or 'poly'")
        if lens is not None:
            raise ValueError("ragged batches are resampled with kind='kaiser_best'")
        if key not in self._resamplers:
            taps, up, down, pre, n_out = audio.design_resampler(sr_in, sr_out, x.shape[1])
            self._resamplers[key] = (torch.from_numpy(taps).to(self.device), up, down, pre, n_out)
        taps, up, down, pre, n_out = self._resamplers[key]
        return self.ctx.resample_poly(x.contiguous(), taps, up, down, pre, n_out)

    def to_int16(self, wave: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """B6 on the GPU (I_ea/predict.py:204-206): fp32 waveform -> int16 PCM, `* 32768` truncated toward zero (si_pcm16)."""
        return self.ctx.pcm16(wave.contiguous(), out)

    def get_mel(self, x: torch.Tensor) -> torch.Tensor:
        """`get_mel(x)` of I_ea/dataset/mel_dump.py:96-98: x (B, n) already normalised -> (B, 80, Tm) log-mel."""
        return self.ctx.mel_frontend(x, None, None, normalize=False)

    def codebook_metrics(self, feats: torch.Tensor, frame_pos: torch.Tensor, lm: int, target: torch.Tensor):
        """Loss half of the reference's cos_sim on the masked frames of `feats` (B, T, 80) against target labels
        (B, Lm): -> dict(loss, loss_terms, pred_labels, cos_pred_target, accuracy)."""
        self._need(codebook=True, what="codebook_metrics")
        loss, terms, pred, cpt = self.ctx.codebook_metrics(feats, frame_pos, lm, target)
        return {"loss": loss[0], "loss_terms": terms, "pred_labels": pred, "cos_pred_target": cpt,
                "accuracy": (pred == target).float().mean()}

    # ---- ragged batches (BASELINE configs[4]): clips of different lengths in ONE set of launches
    def encode_ragged(self, wave16: torch.Tensor, len16, mask_start: Optional[torch.Tensor] = None,
                      mask_len: Optional[torch.Tensor] = None, normalize: bool = True) -> torch.Tensor:
        """wave16 (B, Nmax): clip b = the first len16[b] samples of its row -> (B, Tmax, 80); each clip's frames equal that clip
        encoded alone (the reference handles one file per run, I_ea/predict.py:76-207); rows past a clip's frames are zero."""
        self._need(head=True, what="encode_ragged")
        return self.ctx.hubert_forward_varlen(wave16, len16, mask_start, mask_len, normalize)

    def mel_ragged(self, wave22: torch.Tensor, len22, mask_start: Optional[torch.Tensor] = None,
                   mask_end: Optional[torch.Tensor] = None, normalize: bool = True) -> torch.Tensor:
        """`mel` for clips of different lengths: (B, N22max) + per-clip sample counts -> (B, 80, Tm_max), zero frames past a clip's own."""
        return self.ctx.mel_frontend_varlen(wave22, len22, mask_start, mask_end, normalize)

    def vocode_ragged(self, mel: torch.Tensor, mel_len, stretch: bool = True) -> torch.Tensor:
        return self.ctx.hifigan_forward_varlen(mel, mel_len, stretch)

    def predict_ragged_batch(self, wave16: torch.Tensor, len16, mel: torch.Tensor, mel_len, frame_pos: torch.Tensor, frame_len: int,
                             blind: bool = False, mask_start: Optional[torch.Tensor] = None,
                             mask_len: Optional[torch.Tensor] = None) -> Dict[str, object]:
        """`predict_batch` for clips of DIFFERENT lengths sharing every launch: wave16 (B, Nmax) / mel (B, 80, Tm_max) hold clip b in
        the first len16[b] samples / mel_len[b] frames of its row (host int sequences).  Every clip's outputs equal that clip's
        alone.  -> feats (B, Tmax, 80), labels (B, Lm) (-1 past a clip's own count in blind mode), mel, wave (B, Lmax; zero past a
        clip's own samples), wave_len / frames / mel_len: per-clip valid extents (lists)."""
        self._need(head=True, codebook=True, what="predict_ragged_batch")
        B = wave16.shape[0]
        len16 = [int(n) for n in len16]
        mel_len = [int(n) for n in mel_len]
        frames = [self.ctx.num_frames(n) for n in len16]
        if blind:
            feats = self.encode_ragged(wave16, len16)
            pos = torch.zeros(B, dtype=torch.int32, device=self.device)
            cnt_h = [min(t, m) for t, m in zip(frames, mel_len)]                   # predict_batch's `min(T, Tm)`, per clip
        else:
            if mask_start is None:
                mask_start = frame_pos * 320 + 80                                   # predict.py:133
                mask_len = torch.full_like(frame_pos, max(frame_len * 320 - 81, 0))
            feats = self.encode_ragged(wave16, len16, mask_start.to(torch.int32), mask_len.to(torch.int32))
            pos, cnt_h = frame_pos, [int(frame_len)] * B
        lm = max(cnt_h)
        cnt = torch.tensor(cnt_h, dtype=torch.int32, device=self.device)
        mel2 = mel.clone()
        labels = self.ctx.codebook_splice_varlen(feats, pos, cnt, lm, mel2)
        wav = self.vocode_ragged(mel2, mel_len, stretch=True)
        return {"feats": feats, "labels": labels, "mel": mel2, "wave": wav, "frames": frames, "mel_len": mel_len, "label_cnt": cnt_h,
                "wave_len": [self.ctx.vocoder_samples(m, True) for m in mel_len]}

    def predict_batch(self, wave16: torch.Tensor, mel: torch.Tensor, frame_pos: torch.Tensor, frame_len: int,
                      blind: bool = False, mask_start: Optional[torch.Tensor] = None,
                      mask_len: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """wave16 (B, N) raw 16 kHz clips, mel (B, 80, Tm) log-mel of the masked 22.05 kHz clips, frame_pos (B,) int32
        first masked 20 ms frame, frame_len = Lm.  All tensors on this engine's GPU.  `mel` is not modified.
        blind=True replaces every frame (mask position unknown)."""
        self._need(head=True, codebook=True, what="predict_batch")
        B = wave16.shape[0]
        if blind:
            feats = self.encode(wave16, None, None)
            pos = torch.zeros(B, dtype=torch.int32, device=self.device)
            lm = min(feats.shape[1], mel.shape[2])
        else:
            if mask_start is None:
                mask_start = frame_pos * 320 + 80                                   # predict.py:133
                mask_len = torch.full_like(frame_pos, max(frame_len * 320 - 81, 0))
            feats = self.encode(wave16, mask_start.to(torch.int32), mask_len.to(torch.int32))
            pos, lm = frame_pos, frame_len
        mel2 = mel.clone()
        labels = self.splice(feats, pos, lm, mel2)
        wav = self.vocode(mel2, stretch=True)
        return {"feats": feats, "labels": labels, "mel": mel2, "wave": wav}


# ------------------------------------------------------------------------------------------------------------------
# Drop-in module wrappers (same call signatures as the reference's nn.Modules)
# ------------------------------------------------------------------------------------------------------------------
class CustomModel:
    """`CustomModel.forward(input_values, attention_mask)` (I_ea/model.py:80-89): processor-normalised input_values (B, N)
    -> (B, T, codebook_dim).  attention_mask (B, N) 0/1 marks the real samples of a RIGHT-PADDED batch (what the HF
    processor emits with padding=True): the padded frames are zeroed after the projection and excluded as attention
    keys, as modeling_hubert.py:921-932,428-437 does; the output is defined on all T frames, like the reference's."""

    def __init__(self, engine: InpaintingEngine):
        self.engine = engine

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.engine.device:
            raise RuntimeError("the engine is bound to its GPU at construction")
        return self

    def __call__(self, input_values: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = input_values.to(self.engine.device, torch.float32).contiguous()
        valid = None
        if attention_mask is not None:
            m = attention_mask.to(self.engine.device).bool()
            if not bool(m.all()):
                if m.shape != x.shape or bool((m[:, 1:] & ~m[:, :-1]).any()):
                    raise ValueError("attention_mask must be (B, N) and right-padded (ones, then zeros), as the HF processor emits it")
                valid = m.sum(-1).to(torch.int32).contiguous()
        return self.engine.encode(x, None, None, normalize=False, valid_len=valid)

    forward = __call__


class Generator:
    """`Generator.forward` (I_ea/hifi_gan/models.py:107-123): feats (B, 80, T') -> (B, 1, T' * hop)."""

    def __init__(self, engine: InpaintingEngine):
        self.engine = engine

    def eval(self):
        return self

    def remove_weight_norm(self):
        return None            # folded at load (api.hip)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.engine.device, torch.float32).contiguous()
        return self.engine.vocode(x, stretch=False).unsqueeze(1)

    forward = __call__


class LossFunction:
    """`LossFunction` of I_ea/loss_fn.py over the engine's resident codebook: `cos_sim` (loss + arg-max labels, :29-47),
    `cos_sim_target_labels` (:49-62), plus the centroid splice that follows the call in the script
    (I_ea/predict.py:184-187)."""

    def __init__(self, engine: InpaintingEngine):
        self.engine = engine
        self._last = None

    def cos_sim(self, output: torch.Tensor, labels: torch.Tensor):
        """output (B, Lm, 80) gathered frames, labels (B, Lm) int64 -> (loss scalar tensor, pred_labels (B, Lm)),
        as I_ea/loss_fn.py:29-47 (called at I_ea/predict.py:171)."""
        dev = self.engine.device
        self.engine._need(codebook=True, what="LossFunction.cos_sim")
        out = output.to(dev, torch.float32).contiguous()
        lab = labels.to(dev, torch.int64).contiguous()
        pos = torch.zeros(out.shape[0], dtype=torch.int32, device=dev)
        loss, terms, pred, cpt = self.engine.ctx.codebook_metrics(out, pos, out.shape[1], lab)
        self._last = (pred, lab, cpt)
        return loss[0], pred

    def cos_sim_target_labels(self, pred_labels: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """cos(centred predicted centroid, centred target centroid) per frame, flattened (I_ea/loss_fn.py:49-62).
        Served from the `cos_sim` call that produced `pred_labels` (the script always calls the two back to back,
        I_ea/predict.py:171-173); any other label pair is an error rather than a silent host computation."""
        if self._last is None or not (torch.equal(self._last[0], pred_labels.to(self._last[0].device)) and
                                      torch.equal(self._last[1], labels.to(self._last[1].device))):
            raise ValueError("cos_sim_target_labels expects the (pred_labels, labels) pair of the preceding cos_sim call")
        return self._last[2].reshape(-1)

    def predict_and_splice(self, outputs: torch.Tensor, mask_pos: torch.Tensor, mask_len: int, mel: torch.Tensor):
        """outputs (B, T, 80), mask_pos (B,) int32, mel (B, 80, Tm) modified in place -> labels (B, Lm)."""
        return self.engine.splice(outputs.contiguous(), mask_pos.to(self.engine.device, torch.int32).contiguous(),
                                  int(mask_len), mel)


class CodeGenerator:
    """`CodeGenerator.forward` of I_da/src/model.py:124-189 in its look-up-table configuration (hubert_lut.json: content
    units + quantised F0 + speaker embedding -> 384 channels -> the unit HiFi-GAN, upsample rates 5, 4, 4, 2, 2): the
    embedding / `_upsample` / concat front is si_unit_frontend, the generator is the engine's (a VocoderArch with
    num_mels = 3 * embedding_dim).  `emb_c` / `emb_p` are the checkpoint's `emb_c.weight` / `emb_p.weight` tables.
    The reference quantises F0 with its fixed VQ-VAE inside forward (:160-166): pass `f0_quantizer=F0Quantizer(engine,
    fo_vqvae_state)` and call with `f0=`; or pass the indices directly as `f0_code=`."""

    def __init__(self, engine: InpaintingEngine, emb_c: torch.Tensor, emb_p: Optional[torch.Tensor] = None,
                 f0_quantizer: Optional["F0Quantizer"] = None):
        self.engine = engine
        dev = engine.device
        self.emb_c = emb_c.to(dev, torch.float32).contiguous()
        self.emb_p = None if emb_p is None else emb_p.to(dev, torch.float32).contiguous()
        self.f0_quantizer = f0_quantizer                               # `self.fo_vqvae` of the reference (:63-71)

    def eval(self):
        return self

    def remove_weight_norm(self):
        return None

    def __call__(self, **kwargs) -> torch.Tensor:
        """code (B, Frame) int64; f0 (B, 1, Frame_f0) fp32 as in the reference (quantised here by the fixed F0 VQ-VAE when
        the generator was built with an `F0Quantizer`) or f0_code (B, Frame_p) int64 directly; emb (B, Emb) speaker
        embedding (optional) -> (B, 1, Frame * hop) waveform."""
        dev = self.engine.device
        code = kwargs["code"].to(dev, torch.int64).contiguous()
        f0c = kwargs.get("f0_code")
        if f0c is None and kwargs.get("f0") is not None:
            if self.f0_quantizer is None:
                raise ValueError("CodeGenerator: an F0 track needs the F0 VQ-VAE (pass f0_quantizer=F0Quantizer(...)) or f0_code")
            f0c = self.f0_quantizer(kwargs["f0"])
        emb = kwargs.get("emb")
        x = self.engine.ctx.unit_frontend(code, self.emb_c,
                                          None if f0c is None else f0c.to(dev, torch.int64).contiguous(), self.emb_p,
                                          None if emb is None else emb.to(dev, torch.float32).contiguous())
        return self.engine.vocode(x, stretch=False).unsqueeze(1)

    forward = __call__


class F0Quantizer:
    """The fixed F0 VQ-VAE front that `CodeGenerator.forward` runs on the F0 track (I_da/src/model.py:160-163):
    `z_p = fo_vqvae.vq(fo_vqvae.encoder(fo))[0][0]` -- jukebox.py `Encoder` (one level) on the GPU (si_f0_encoder_forward),
    then the bottleneck's nearest-codebook arg-min (vq.py:117-127; si_kmeans_assign).  `state` is the `FoVQVAE`
    state dict (`encoder.level_blocks.0.model...`, `vq.level_blocks.0.k`)."""

    def __init__(self, engine: InpaintingEngine, state: dict, desc: Optional["native.F0EncDesc"] = None):
        from . import native
        self.engine = engine
        self.desc = desc or native.F0EncDesc()
        dev = engine.device
        self.weights = pack_f0_encoder(state, self.desc).to(dev)
        self.codebook = state["vq.level_blocks.0.k"].to(dev, torch.float32).contiguous()

    def features(self, f0: torch.Tensor) -> torch.Tensor:
        """f0 (B, 1, T) -> (B, T', 128) encoder output, channels-last."""
        return self.engine.ctx.f0_encoder(self.desc, self.weights, f0.to(self.engine.device, torch.float32).contiguous())

    def codes(self, h: torch.Tensor) -> torch.Tensor:
        """encoder output (B, T', 128) -> z_p (B, T') int64: the bottleneck's nearest-codebook arg-min (vq.py:117-127)."""
        B, Tp, E = h.shape
        return self.engine.ctx.kmeans_assign(h.reshape(B * Tp, E), self.codebook).reshape(B, Tp)

    def __call__(self, f0: torch.Tensor) -> torch.Tensor:
        """f0 (B, 1, T) -> z_p (B, T') int64, the indices `emb_p` is looked up with."""
        return self.codes(self.features(f0))


def pack_f0_encoder(state: dict, desc) -> torch.Tensor:
    """Flatten the encoder's parameters in module order (the order si_f0_encoder_forward consumes them):
    `encoder.level_blocks.0.model.<i>.0` = strided conv, `.model.<i>.1.model.<j>.model.1` / `.model.3` = the res block's
    k3 / k1 convs (resnet.py:37-42), `.model.<down_t>` = the last conv (jukebox.py:80-82)."""
    pre = "encoder.level_blocks.0.model."
    parts = []
    for i in range(desc.down_t):
        parts += [state[f"{pre}{i}.0.weight"], state[f"{pre}{i}.0.bias"]]
        for j in range(desc.depth):
            r = f"{pre}{i}.1.model.{j}.model."
            parts += [state[r + "1.weight"], state[r + "1.bias"], state[r + "3.weight"], state[r + "3.bias"]]
    parts += [state[f"{pre}{desc.down_t}.weight"], state[f"{pre}{desc.down_t}.bias"]]
    return torch.cat([p.detach().to(torch.float32).reshape(-1) for p in parts]).contiguous()


class Metrics:
    """The signal metrics of `Metrics` (I_ea/metrics.py:12-142) on the GPU: `avg_cosine_sim`, `avg_d2_dist`, `rmse` on mel
    segments (80, L) and `sisdr` on waveforms, same names and argument order.  The Whisper WER/CER and PESQ/STOI members
    are third-party CPU code outside the hot path and are not provided.  `centroids` is what the
    reference passes to the constructor: the (80,) codebook mean that `avg_cosine_sim` subtracts."""

    def __init__(self, engine: InpaintingEngine, centroids: torch.Tensor):
        self.engine = engine
        self.center = centroids.reshape(-1).to(engine.device, torch.float32).contiguous()

    def _pair(self, t1, t2):
        dev = self.engine.device
        return t1.to(dev, torch.float32).contiguous()[None], t2.to(dev, torch.float32).contiguous()[None]

    def avg_cosine_sim(self, tensor1: torch.Tensor, tensor2: torch.Tensor) -> torch.Tensor:
        a, b = self._pair(tensor1, tensor2)
        return self.engine.ctx.mel_metrics(a, b, self.center)[0, 0]

    def avg_d2_dist(self, tensor1: torch.Tensor, tensor2: torch.Tensor) -> torch.Tensor:
        a, b = self._pair(tensor1, tensor2)
        return self.engine.ctx.mel_metrics(a, b, None)[0, 1]

    def rmse(self, tensor1: torch.Tensor, tensor2: torch.Tensor) -> torch.Tensor:
        a, b = self._pair(tensor1, tensor2)
        return self.engine.ctx.mel_metrics(a, b, None)[0, 2]

    def sisdr(self, x_est, x_ref) -> float:
        dev = self.engine.device
        e = torch.as_tensor(x_est, dtype=torch.float32).reshape(1, -1).to(dev).contiguous()
        r = torch.as_tensor(x_ref, dtype=torch.float32).reshape(1, -1).to(dev).contiguous()
        return float(self.engine.ctx.sisdr(e, r)[0])
