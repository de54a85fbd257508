"""Host-side audio glue around the hot path: wav I/O and resampling (still host code).

The masking / peak normalisation / log-mel front-end of the 22.05 kHz side (row f-1) is in the HIP library
(`si_mel_frontend`, csrc/frontend_kernels.hip); nothing here computes a mel.

* resampling: `librosa.load(sr=...)` (I_ea/predict.py:79-80) uses a Kaiser-windowed sinc resampler; here scipy's
  polyphase resampler with a Kaiser window.  Not bit-identical to librosa; documented as such (librosa is absent, so
  there is no in-container oracle for it).
"""
from __future__ import annotations

import math
import numpy as np
import torch

MAX_WAV_VALUE = 32768.0


def read_wav(path: str):
    """-> (float32 mono in [-1, 1], sample_rate)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    return x, int(sr)


KAISER_BETA = 14.769656459379492


def design_resampler(sr_in: int, sr_out: int, n_in: int):
    """Filter and index bookkeeping of `scipy.signal.resample_poly(x, up, down, window=("kaiser", beta))` for the GPU
    resampler (si_resample_poly): returns (taps float32 incl. resample_poly's zero pre-padding, up, down, pre_remove,
    n_out).  The filter is designed in float64 exactly as resample_poly designs it (firwin(2 * 10 * max(up, down) + 1,
    1 / max(up, down)) * up) and rounded to float32 once."""
    from scipy.signal import firwin
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    half_len = 10 * max(up, down)
    h = firwin(2 * half_len + 1, 1.0 / max(up, down), window=("kaiser", KAISER_BETA)) * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    taps = np.concatenate([np.zeros(n_pre_pad), h]).astype(np.float32)
    return taps, up, down, n_pre_remove, n_out


KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=KAISER_BETA)


def design_kaiser_best(sr_in: int, sr_out: int, n_in: int):
    """Tables of resampy's `kaiser_best` band-limited interpolation -- the resampler behind `librosa.load(path, sr=...)` in the
    reference's pinned librosa==0.9.1 (I_ea/predict.py:79-80; requirements.txt:3) -- for the GPU kernel (si_resample_sinc):
    win = rolloff * sinc(rolloff * t), t in [0, 64] at 512 samples per zero crossing, tapered by the right half of a Kaiser window
    (beta 14.77), scaled by the ratio when down-sampling; dwin = its forward differences; time_reg[t] = output sample t's input time,
    1 / ratio accumulated by REPEATED ADDITION in float64 as resampy's loop does (at exactly-integer times the accumulated rounding
    selects the table phase, and the phases differ because the table step is truncated to an integer).
    -> dict(win, dwin, time_reg float64 numpy; num_table, step, scale, ratio, n_out = ceil(n_in * ratio) (librosa's fix_length))."""
    from scipy.signal.windows import kaiser
    ratio = float(sr_out) / float(sr_in)
    nb = 2 ** KAISER_BEST["precision"]
    n = nb * KAISER_BEST["num_zeros"]
    r = KAISER_BEST["rolloff"]
    win = kaiser(2 * n + 1, KAISER_BEST["beta"])[n:] * (r * np.sinc(r * np.linspace(0, KAISER_BEST["num_zeros"], num=n + 1, endpoint=True)))
    if ratio < 1:
        win = win * ratio
    dwin = np.zeros_like(win)
    dwin[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    n_out = int(math.ceil(n_in * ratio))
    inc = 1.0 / ratio
    tr = np.cumsum(np.full(max(n_out, 1), inc)) - inc
    tr[0] = 0.0
    return dict(win=np.ascontiguousarray(win), dwin=dwin, time_reg=tr[:n_out].copy(), num_table=nb, step=int(scale * nb), scale=scale,
                ratio=ratio, n_out=n_out)


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """Host (scipy) resampler with the same filter: the reference of the GPU one in the tests; predict.py resamples on
    the GPU (InpaintingEngine.resample)."""
    if sr_in == sr_out:
        return x.astype(np.float32)
    from scipy.signal import resample_poly
    g = math.gcd(sr_in, sr_out)
    return resample_poly(x.astype(np.float64), sr_out // g, sr_in // g, window=("kaiser", KAISER_BETA)).astype(np.float32)


def load_audio(path: str, sr: int) -> np.ndarray:
    """Stand-in for `librosa.load(path, sr=sr)` (I_ea/predict.py:79-80)."""
    x, sr_in = read_wav(path)
    return resample(x, sr_in, sr)


def to_int16_pcm(audio: torch.Tensor, clip: bool = False) -> np.ndarray:
    """B6, the script's own two statements (I_ea/predict.py:204-206): `audio * 32768` in fp32, then numpy
    `.astype('int16')` -- truncation toward zero, no rounding.  The generator ends in tanh, so the product lies in
    [-32768, 32768]; only an exact +1.0 sample (tanhf saturates for pre-activations above ~9) is out of int16's range, and a
    float -> int16 cast of 32768.0 is undefined behaviour in C and numpy (x86 scalar code gives -32768: a full-scale negative
    click).  That ONE value is pinned to 32767; everywhere the reference's cast is defined the PCM is bit-identical to it.  The
    GPU form of the same statement is `InpaintingEngine.to_int16` (si_pcm16).
    clip=True is an opt-in deviation for callers who feed waveforms that did not come out of the generator: values are
    clamped to [-32768, 32767] before the cast."""
    a = audio.detach().float().cpu() * MAX_WAV_VALUE
    a = a.clamp(-32768.0, 32767.0) if clip else a.clamp(max=32767.0)
    return a.numpy().astype("int16")


def write_wav(path: str, pcm_or_float: np.ndarray, sr: int):
    from scipy.io import wavfile
    wavfile.write(path, sr, pcm_or_float)
