"""`predict.py` entry point of the I_ea path, re-stated over the HIP engine.

Run as the reference is run (I_ea/predict.py:58-62): from a directory holding `predict.yaml`

    python -m speech_inpainting_amd.predict            # or: python predict.py

It reads the reference's YAML schema, loads the same three artefacts (CustomModel `.pt` / local HF directory, HiFi-GAN
`generator` checkpoint + `config.json`, joblib k-means codebook), and writes the reference's output files
`<save_pred>/<wave_name>/{orig,masked,hifi_masked,expected_inpaint,inpainted}.wav` (I_ea/predict.py:84,128,134,201,207;
`expected_inpaint.wav` only when the ground-truth label file exists).  `predict_clips` is the importable batch form.

Differences from the script, all outside the three replaced subsystems: no Whisper `Metrics` object is built (the
script constructs it and never uses it, I_ea/predict.py:72-73), PNG plots are skipped, and the two `librosa.load` calls are one
wav read plus the resampler librosa 0.9.1 itself uses (resampy `kaiser_best`) run on the GPU (si_resample_sinc; pinned against the
reference-held LJ001-0001 22k / 16k pair), the clips staying on the device from there to the int16 conversion (si_pcm16).
"""
from __future__ import annotations

import os
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import audio
from .arch import HubertArch, VocoderArch
from .checkpoint import arch_for_type, load_codebook, load_generator_checkpoint, load_hubert_checkpoint
from .config import PredictConfig, choose_device, load_predict_config
from .engine import InpaintingEngine


def build_engine(cfg: PredictConfig, device: Optional[torch.device] = None, encoder_dtype: str = "fp32",
                 vocoder_dtype: str = "fp32") -> InpaintingEngine:
    device = device or choose_device(cfg.device_index)
    if device.type != "cuda":
        raise RuntimeError("no GPU selected/available: this package has no CPU path (device.index in the YAML)")
    hsd, harch = load_hubert_checkpoint(cfg.hubert_checkpoint, cfg.hubert_type)
    harch = harch or arch_for_type(cfg.hubert_type)
    gsd, varch = load_generator_checkpoint(cfg.hifigan_checkpoint)
    cb = load_codebook(cfg.km_model_path)
    if cb.shape[0] != cfg.n_clusters:
        raise ValueError(f"{cfg.km_model_path}: {cb.shape[0]} centroids but km_model.n_clusters = {cfg.n_clusters}")
    eng = InpaintingEngine(harch, varch, cfg.n_clusters, device, encoder_dtype, vocoder_dtype)
    return eng.load_state(hsd, gsd, cb)


def check_mask_span(engine: InpaintingEngine, n16: int, n22: int, mask_pos: Sequence[int], mask_frames: int) -> None:
    """The masked frames must exist on both sides: [pos, pos + Lm) inside the T encoder frames AND the Tm mel frames
    (for a 4 s clip T = 199 but Tm = 200).  The reference fails on the slice-shape mismatch at I_ea/predict.py:166-168,
    185-187; the kernels would leave such frames unspliced (label -1), so refuse here where positions are host ints."""
    T, Tm = engine.ctx.num_frames(n16), engine.ctx.mel_frames(n22)
    for i, p in enumerate(mask_pos):
        if int(p) < 0 or int(p) + int(mask_frames) > min(T, Tm):
            raise ValueError(f"clip {i}: masked frames [{int(p)}, {int(p) + int(mask_frames)}) do not fit the clip "
                             f"({T} encoder frames, {Tm} mel frames)")


def predict_clips(engine: InpaintingEngine, waves16: Sequence[np.ndarray], waves22: Sequence[np.ndarray],
                  mask_pos: Sequence[int], mask_frames: int, blind: bool = False,
                  mask22: Optional[Sequence[Tuple[int, int]]] = None, diagnostics: bool = False,
                  target_labels: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Batch form of I_ea/predict.py:97-207 for clips of EQUAL length.
    waves16 / waves22: the same clips at 16 kHz / 22.05 kHz (float32, un-normalised), mask_pos: first masked 20 ms frame.
    mask22: per-clip [start, end) of the span zeroed on the 22.05 kHz side (predict.py:99-102: the 16 kHz sample
    positions of the YAML times scaled by 22050 // 16000); default = the frame span.
    diagnostics=True adds the script's other two vocoder passes as batch outputs: `hifi_masked` (the generator on the
    masked mel alone, predict.py:123-128) and -- when `target_labels` (B, Lm) int64 ground-truth codewords are given --
    `expected_inpaint` (their centroids spliced instead of the predicted ones, predict.py:177-189,198-201) plus the
    codeword metrics of predict.py:171-173 (`loss`, `cos_pred_target`)."""
    dev = engine.device
    n16, n22 = len(waves16[0]), len(waves22[0])
    if any(len(w) != n16 for w in waves16) or any(len(w) != n22 for w in waves22):
        raise ValueError("clips in one batch must have equal length (predict_clips_ragged / predict_ragged take clips of "
                         "different lengths)")
    wave22 = torch.from_numpy(np.stack([np.asarray(w, dtype=np.float32) for w in waves22])).to(dev)
    wave = torch.from_numpy(np.stack([np.asarray(w, dtype=np.float32) for w in waves16])).to(dev)
    return predict_resident(engine, wave, wave22, mask_pos, mask_frames, blind, mask22, diagnostics, target_labels)


def predict_resident(engine: InpaintingEngine, wave: torch.Tensor, wave22: torch.Tensor, mask_pos: Sequence[int], mask_frames: int,
                     blind: bool = False, mask22: Optional[Sequence[Tuple[int, int]]] = None, diagnostics: bool = False,
                     target_labels: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """`predict_clips` on clips that are already on the GPU: wave (B, n16) / wave22 (B, n22) float32 device tensors (e.g. straight
    out of `engine.resample`); same outputs."""
    dev = engine.device
    n16, n22 = wave.shape[1], wave22.shape[1]
    if not blind:
        check_mask_span(engine, n16, n22, mask_pos, mask_frames)
    if blind:
        mel = engine.mel(wave22)                                                    # nothing zeroed; predict.py:104-106
    else:
        if mask22 is None:
            mask22 = [(p * 320 * 22050 // 16000, (p + mask_frames) * 320 * 22050 // 16000) for p in mask_pos]
        s22 = torch.tensor([min(max(int(a), 0), n22) for a, _ in mask22], dtype=torch.int32, device=dev)
        e22 = torch.tensor([min(max(int(b), 0), n22) for _, b in mask22], dtype=torch.int32, device=dev)
        mel = engine.mel(wave22, s22, e22)                                          # predict.py:99-106 on the GPU
    pos = torch.tensor(list(mask_pos), dtype=torch.int32, device=dev)
    if diagnostics and not blind:
        # The script's three generator passes (masked / expected / inpainted mel, I_ea/predict.py:123-128,196-207) differ only in the
        # Lm spliced frames: ONE full pass (the masked mel), the other two over the window the spliced frames can reach
        # (engine.vocode_window: bit-identical to full passes).
        feats = engine.encode(wave, (pos * 320 + 80).to(torch.int32), torch.full_like(pos, max(mask_frames * 320 - 81, 0)))
        mel2 = mel.clone()
        labels = engine.splice(feats, pos, mask_frames, mel2)
        base = engine.vocode(mel, stretch=True)
        out = {"feats": feats, "labels": labels, "mel": mel2, "hifi_masked": base,
               "wave": engine.vocode_window(base, mel2, list(mask_pos), mask_frames)}
    else:
        out = engine.predict_batch(wave, mel, pos, mask_frames, blind=blind)
        if diagnostics:
            out["hifi_masked"] = engine.vocode(mel, stretch=True)
    out["mel_masked"] = mel
    if diagnostics:
        if target_labels is not None and not blind:
            tgt = target_labels.to(dev, torch.int64).contiguous()
            exp = mel.clone()
            engine.splice_labels(tgt, pos, exp)
            out["expected_inpaint"] = engine.vocode_window(out["hifi_masked"], exp, list(mask_pos), mask_frames)
            m = engine.codebook_metrics(out["feats"], pos, mask_frames, tgt)
            out["loss"], out["cos_pred_target"] = m["loss"], m["cos_pred_target"]
    return out


def bucket_by_length(lengths: Sequence[int], max_batch: int = 32) -> List[List[int]]:
    """Group clip indices into batches of EXACTLY equal length (at most `max_batch` each), longest first, original
    order kept inside a group.  HuBERT-base's GroupNorm over time and the processor's per-clip normalisation make
    padded batches differ from the reference's one-clip-at-a-time loop (I_ea/predict.py:76-207 handles one file),
    so ragged inputs are bucketed instead of padded."""
    groups: Dict[int, List[int]] = {}
    for i, n in enumerate(lengths):
        groups.setdefault(int(n), []).append(i)
    out: List[List[int]] = []
    for n in sorted(groups, reverse=True):
        idx = groups[n]
        out.extend(idx[k:k + max_batch] for k in range(0, len(idx), max_batch))
    return out


def plan_ragged_batches(lengths: Sequence[int], max_batch: int = 32, max_waste: Optional[float] = None) -> List[List[int]]:
    """Clip indices sorted by length (longest first) and cut into batches of at most `max_batch`; with `max_waste` a batch is also
    closed before its STORAGE padding 1 - sum(len) / (n * longest) would exceed it.  (The kernels number their tiles clip by clip
    without gaps, so padding costs memory, not launched work: one full batch fills the chip best.)"""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    out: List[List[int]] = []
    cur: List[int] = []
    tot = 0
    for i in order:
        n = int(lengths[i])
        if cur:
            longest = int(lengths[cur[0]])
            waste = 1.0 - (tot + n) / ((len(cur) + 1) * longest)
            if len(cur) >= max_batch or (max_waste is not None and waste > max_waste):
                out.append(cur)
                cur, tot = [], 0
        cur.append(i)
        tot += n
    if cur:
        out.append(cur)
    return out


def storage_padding(lengths: Sequence[int], batches: Sequence[Sequence[int]]) -> float:
    """Fraction of the padded (batch, longest) storage that holds no sample, over all batches."""
    real = sum(int(lengths[i]) for b in batches for i in b)
    padded = sum(len(b) * max(int(lengths[i]) for i in b) for b in batches)
    return 1.0 - real / max(padded, 1)


def pad_stack(waves: Sequence[np.ndarray]) -> Tuple[torch.Tensor, List[int]]:
    """Clips of different lengths -> pinned-memory-free (B, longest) float32 host tensor (zero tail) + their lengths."""
    lens = [len(w) for w in waves]
    out = torch.zeros(len(waves), max(lens), dtype=torch.float32)
    for i, w in enumerate(waves):
        out[i, :lens[i]] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out, lens


def predict_clips_ragged(engine: InpaintingEngine, waves16: Sequence[np.ndarray], waves22: Sequence[np.ndarray],
                         mask_pos: Sequence[int], mask_frames: int, blind: bool = False,
                         mask22: Optional[Sequence[Tuple[int, int]]] = None) -> Dict[str, object]:
    """`predict_clips` for clips of DIFFERENT lengths in ONE set of launches (the library's ragged-batch entry points): every
    clip's outputs equal that clip's alone.  Tensors are (B, longest ...); `wave_len`, `frames`, `mel_len` give each clip's extent."""
    dev = engine.device
    if not blind:
        for i, (a, b) in enumerate(zip(waves16, waves22)):
            check_mask_span(engine, len(a), len(b), [mask_pos[i]], mask_frames)
    w22, len22 = pad_stack(waves22)
    w16, len16 = pad_stack(waves16)
    wave22 = w22.to(dev)
    if blind:
        mel = engine.mel_ragged(wave22, len22)
    else:
        if mask22 is None:
            mask22 = [(p * 320 * 22050 // 16000, (p + mask_frames) * 320 * 22050 // 16000) for p in mask_pos]
        s22 = torch.tensor([min(max(int(a), 0), n) for (a, _), n in zip(mask22, len22)], dtype=torch.int32, device=dev)
        e22 = torch.tensor([min(max(int(b), 0), n) for (_, b), n in zip(mask22, len22)], dtype=torch.int32, device=dev)
        mel = engine.mel_ragged(wave22, len22, s22, e22)
    mel_len = [engine.ctx.mel_frames(n) for n in len22]
    pos = torch.tensor(list(mask_pos), dtype=torch.int32, device=dev)
    out = engine.predict_ragged_batch(w16.to(dev), len16, mel, mel_len, pos, mask_frames, blind=blind)
    out["mel_masked"] = mel
    return out


def predict_ragged(engine: InpaintingEngine, waves16: Sequence[np.ndarray], waves22: Sequence[np.ndarray],
                   mask_pos: Sequence[int], mask_frames: int, blind: bool = False, max_batch: int = 32,
                   max_waste: Optional[float] = None, exact_length: bool = False) -> List[Dict[str, torch.Tensor]]:
    """The path over clips of DIFFERENT lengths (BASELINE configs[4]; the reference runs one file of any length per invocation,
    I_ea/predict.py:76-207): clips are sorted by length and cut into ragged batches (`plan_ragged_batches`) that share every
    launch; each clip's result equals that clip run alone.  Returns one dict per input clip, in input order, cut to the clip's own
    extent (tensors keep a batch dimension of 1).  exact_length=True is the older route -- batches of EXACTLY equal length through
    the uniform entry points (singletons on continuous lengths) -- kept as the reference the ragged route is tested against."""
    if not (len(waves16) == len(waves22) == len(mask_pos)):
        raise ValueError("waves16, waves22 and mask_pos must have one entry per clip")
    results: List[Optional[Dict[str, torch.Tensor]]] = [None] * len(waves16)
    if exact_length:
        # a bucket must agree on BOTH sample counts (the 22.05 kHz length follows from the resampler's rounding)
        keys = [len(a) * 1_000_003 + len(b) for a, b in zip(waves16, waves22)]
        for idx in bucket_by_length(keys, max_batch):
            out = predict_clips(engine, [waves16[i] for i in idx], [waves22[i] for i in idx], [mask_pos[i] for i in idx],
                                mask_frames, blind=blind)
            for k, i in enumerate(idx):
                results[i] = {name: v[k:k + 1] for name, v in out.items()}
        return results  # type: ignore[return-value]
    for idx in plan_ragged_batches([len(w) for w in waves16], max_batch, max_waste):
        out = predict_clips_ragged(engine, [waves16[i] for i in idx], [waves22[i] for i in idx], [mask_pos[i] for i in idx],
                                   mask_frames, blind=blind)
        for k, i in enumerate(idx):
            T, Tm, nl, nw = out["frames"][k], out["mel_len"][k], out["label_cnt"][k], out["wave_len"][k]
            results[i] = {"feats": out["feats"][k:k + 1, :T], "labels": out["labels"][k:k + 1, :nl], "mel": out["mel"][k:k + 1, :, :Tm],
                          "mel_masked": out["mel_masked"][k:k + 1, :, :Tm], "wave": out["wave"][k:k + 1, :nw]}
    return results  # type: ignore[return-value]


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = load_predict_config(argv[0] if argv else "predict.yaml")
    device = choose_device(cfg.device_index)
    print("Current device:", device)
    engine = build_engine(cfg, device)
    wave_name = cfg.wave_path.split("/")[-1].split(".")[0]
    save_dir = os.path.join(cfg.save_pred, wave_name)
    os.makedirs(save_dir, exist_ok=True)
    raw, sr_file = audio.read_wav(cfg.wave_path)                                   # predict.py:79-80 (librosa.load x 2): one read,
    raw_dev = torch.from_numpy(raw)[None].to(engine.device)                        # both rates on the GPU with librosa 0.9.1's own
    wave_22 = engine.resample(raw_dev, sr_file, 22050)                             # resampler (resampy kaiser_best); they STAY there
    wave_16 = engine.resample(raw_dev, sr_file, 16000)
    audio.write_wav(os.path.join(save_dir, "orig.wav"), wave_16[0].cpu().numpy(), 16000)
    pos, lm = cfg.mask_pos, cfg.mask_frames
    masked_16 = wave_16[0].clone()
    masked_16[pos * 320 + 80:(pos + lm) * 320 + 79 - 80] = 0                       # predict.py:133
    audio.write_wav(os.path.join(save_dir, "masked.wav"), masked_16.cpu().numpy(), 16000)

    span22 = (cfg.start_sample * 22050 // 16000, cfg.end_sample * 22050 // 16000)   # predict.py:99-100
    labels_path = os.path.join(cfg.path2centroids, wave_name + "_labels.pt")
    labels = None
    if os.path.exists(labels_path):                                                # predict.py:160-161
        labels = torch.load(labels_path, map_location="cpu").t().reshape(-1)[pos:pos + lm].long()
    out = predict_resident(engine, wave_16, wave_22, [pos], lm, mask22=[span22], diagnostics=True,
                           target_labels=None if labels is None else labels[None])
    pcm = lambda w: engine.to_int16(w[0]).cpu().numpy()                            # predict.py:204-206 on the GPU (si_pcm16)
    # hifi_masked.wav: the vocoder on the masked mel alone (predict.py:123-128)
    audio.write_wav(os.path.join(save_dir, "hifi_masked.wav"), pcm(out["hifi_masked"]), 22050)
    if labels is not None:                                                         # predict.py:171-189,198-201
        audio.write_wav(os.path.join(save_dir, "expected_inpaint.wav"), pcm(out["expected_inpaint"]), 22050)
        print("Loss:", float(out["loss"]))
        print("Average Cosine Similarity: ", float(out["cos_pred_target"].mean()))
        print("Target codewords: ", labels.tolist())
    print("Predicted codewords: ", out["labels"][0].tolist())
    audio.write_wav(os.path.join(save_dir, "inpainted.wav"), pcm(out["wave"]), 22050)
    print("wrote", save_dir)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
