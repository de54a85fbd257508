"""Request front of the path: everything `I_ea/predict.py` does around the three replaced
subsystems for ONE file -- `librosa.load` at both rates (:79-80), masking + normalisation + log-mel (:99-106,132-141), the model
calls, `audio * 32768` + int16 (:204-206) -- as a pipelined batch service on the GPU:

    host clips (file rate, float32) --pinned H2D (its own stream)--> resample to 22.05 / 16 kHz (resampy kaiser_best, si_resample_sinc)
        -> masked log-mel (si_mel_frontend) -> encoder -> arg-max / splice -> vocoder -> int16 PCM (si_pcm16)
        --async D2H (its own stream)--> pinned host PCM

Batches are double-buffered over two copy streams: while batch i computes, batch i + 1's clips cross PCIe and batch i - 1's PCM comes back, so the
end-to-end rate is the compute rate as long as PCIe is the shorter leg (32 x 4 s: 11 MB in, 5.6 MB out per 12 ms step).
Clips of one batch may have different lengths: they then share every launch through the library's ragged-batch entry points.
PyTorch owns the streams, events and pinned buffers; all arithmetic is the HIP library's.  No CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .engine import InpaintingEngine


@dataclass
class Request:
    """One batch of clips at the FILE's sample rate (what `sf.read` / `librosa.load(sr=None)` yields: float32 in [-1, 1])."""
    clips: Sequence[np.ndarray]
    mask_pos: Sequence[int]                 # first masked 20 ms frame per clip (ignored when blind)
    mask_frames: int = 10
    blind: bool = False
    tag: object = None


@dataclass
class Result:
    pcm: List[np.ndarray]                   # int16 at 22.05 kHz, one array per clip, cut to the clip's own length
    labels: torch.Tensor                    # (B, Lm) predicted codewords (host)
    tag: object = None


@dataclass
class _Slot:
    pin_in: Optional[torch.Tensor] = None
    dev_in: Optional[torch.Tensor] = None
    pin_out: Optional[torch.Tensor] = None
    pin_lab: Optional[torch.Tensor] = None
    ev_h2d: torch.cuda.Event = field(default_factory=torch.cuda.Event)
    ev_done: torch.cuda.Event = field(default_factory=torch.cuda.Event)
    ev_d2h: torch.cuda.Event = field(default_factory=torch.cuda.Event)
    keep: tuple = ()                        # device tensors of the batch in flight (alive until its PCM is back)
    meta: Optional[dict] = None


class RequestFront:
    """Pipelined `predict` service over one engine.  `run(requests)` yields one Result per Request, in order."""

    def __init__(self, engine: InpaintingEngine, sr_in: int = 22050, depth: int = 2):
        self.engine, self.sr_in, self.depth = engine, int(sr_in), max(1, int(depth))
        self.dev = engine.device
        # two copy streams: batch i + 1's clips must not queue behind batch i's PCM, which waits for batch i's compute
        self.h2d = torch.cuda.Stream(self.dev)
        self.d2h = torch.cuda.Stream(self.dev)
        self.slots = [_Slot() for _ in range(self.depth)]

    # ---- sizes of librosa.load's outputs (librosa.resample: ceil(n * ratio) after fix_length)
    def _len_at(self, n: int, sr: int) -> int:
        return n if sr == self.sr_in else int(math.ceil(n * float(sr) / self.sr_in))

    def _ensure(self, s: _Slot, B: int, n_in: int, n_out: int, lm: int):
        # flat, persistent buffers per slot: the device side of the H2D copy must never be a block the caching allocator could
        # hand out while kernels of the compute stream still use it (the copy stream is not ordered with that stream's frees)
        if s.pin_in is None or s.pin_in.numel() < B * n_in:
            s.pin_in = torch.empty(B * n_in, dtype=torch.float32).pin_memory()
            s.dev_in = torch.empty(B * n_in, dtype=torch.float32, device=self.dev)
        if s.pin_out is None or s.pin_out.numel() < B * n_out:
            s.pin_out = torch.empty(B * n_out, dtype=torch.int16).pin_memory()
        if s.pin_lab is None or s.pin_lab.numel() < B * max(lm, 1):
            s.pin_lab = torch.empty(B * max(lm, 1), dtype=torch.int64).pin_memory()

    def _submit(self, s: _Slot, rq: Request):
        eng, dev = self.engine, self.dev
        B = len(rq.clips)
        lens = [len(c) for c in rq.clips]
        n_in = max(lens)
        ragged = min(lens) != n_in
        len22 = [self._len_at(n, 22050) for n in lens]
        len16 = [self._len_at(n, 16000) for n in lens]
        mel_len = [eng.ctx.mel_frames(n) for n in len22]
        frames = [eng.ctx.num_frames(n) for n in len16]
        lm = max(min(t, m) for t, m in zip(frames, mel_len)) if rq.blind else int(rq.mask_frames)
        n_wave = eng.ctx.vocoder_samples(max(mel_len), True)
        self._ensure(s, B, n_in, n_wave, lm)
        # host -> pinned (zero tail for shorter clips), pinned -> device on the copy stream
        # (numpy views of the pinned buffers: plain memcpy.  torch CPU ops here would each wake the intra-op thread pool, whose
        #  spinning workers burn the job's CPU quota until the cgroup is throttled -- measured: 70-280 ms stalls per request)
        pin = s.pin_in[:B * n_in].view(B, n_in)
        pin_np = pin.numpy()
        for i, c in enumerate(rq.clips):
            np.copyto(pin_np[i, :lens[i]], np.asarray(c, dtype=np.float32))
            if lens[i] < n_in:
                pin_np[i, lens[i]:] = 0.0
        compute = torch.cuda.current_stream(dev)
        raw = s.dev_in[:B * n_in].view(B, n_in)
        with torch.cuda.stream(self.h2d):
            raw.copy_(pin, non_blocking=True)
            s.ev_h2d.record(self.h2d)
        compute.wait_event(s.ev_h2d)
        # librosa.load x 2 (I_ea/predict.py:79-80): the file's samples at 22.05 kHz and at 16 kHz
        w22 = raw if self.sr_in == 22050 else eng.resample(raw, self.sr_in, 22050, lens=lens if ragged else None)
        w16 = raw if self.sr_in == 16000 else eng.resample(raw, self.sr_in, 16000, lens=lens if ragged else None)
        pos = torch.tensor([int(p) for p in rq.mask_pos], dtype=torch.int32).to(dev, non_blocking=True)
        if rq.blind:
            s22 = e22 = None
        else:
            s22 = torch.tensor([min(int(p) * 320 * 22050 // 16000, n) for p, n in zip(rq.mask_pos, len22)], dtype=torch.int32).to(dev)
            e22 = torch.tensor([min((int(p) + lm) * 320 * 22050 // 16000, n) for p, n in zip(rq.mask_pos, len22)], dtype=torch.int32).to(dev)
        if ragged:
            mel = eng.mel_ragged(w22, len22, s22, e22)
            out = eng.predict_ragged_batch(w16, len16, mel, mel_len, pos, lm, blind=rq.blind)
            wave_len = out["wave_len"]
        else:
            mel = eng.mel(w22, s22, e22)
            out = eng.predict_batch(w16, mel, pos, lm, blind=rq.blind)
            wave_len = [out["wave"].shape[1]] * B
        pcm = eng.to_int16(out["wave"])                              # B6 on the device
        s.ev_done.record(compute)
        self.d2h.wait_event(s.ev_done)
        n_lab = out["labels"].shape[1]
        with torch.cuda.stream(self.d2h):
            s.pin_out[:pcm.numel()].view(pcm.shape).copy_(pcm, non_blocking=True)
            s.pin_lab[:B * n_lab].view(B, n_lab).copy_(out["labels"], non_blocking=True)
            s.ev_d2h.record(self.d2h)
        s.keep = (w22, w16, mel, out, pcm, pos, s22, e22)              # alive until the copies have run
        s.meta = dict(B=B, wave_len=wave_len, n_lab=n_lab, n_wave=pcm.shape[1], tag=rq.tag)

    def _collect(self, s: _Slot) -> Result:
        s.ev_d2h.synchronize()
        m = s.meta
        po = s.pin_out.numpy()[:m["B"] * m["n_wave"]].reshape(m["B"], m["n_wave"])
        pcm = [po[i, :m["wave_len"][i]].copy() for i in range(m["B"])]
        lab = torch.from_numpy(s.pin_lab.numpy()[:m["B"] * m["n_lab"]].reshape(m["B"], m["n_lab"]).copy())
        s.keep, s.meta = (), None
        return Result(pcm, lab, m["tag"])

    def run(self, requests: Iterable[Request]) -> Iterator[Result]:
        inflight: List[_Slot] = []
        for i, rq in enumerate(requests):
            s = self.slots[i % self.depth]
            if s.meta is not None:                                   # the slot's previous batch: its PCM must be out before reuse
                inflight.remove(s)
                yield self._collect(s)
            self._submit(s, rq)
            inflight.append(s)
        for s in inflight:
            yield self._collect(s)


def predict_stream(engine: InpaintingEngine, requests: Iterable[Request], sr_in: int = 22050, depth: int = 2) -> Iterator[Result]:
    """`predict` over a stream of batches with transfers overlapped (see RequestFront)."""
    return RequestFront(engine, sr_in, depth).run(requests)
