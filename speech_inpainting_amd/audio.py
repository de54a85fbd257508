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


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    if sr_in == sr_out:
        return x.astype(np.float32)
    from scipy.signal import resample_poly
    g = math.gcd(sr_in, sr_out)
    return resample_poly(x.astype(np.float64), sr_out // g, sr_in // g, window=("kaiser", 14.769656459379492)).astype(np.float32)


def load_audio(path: str, sr: int) -> np.ndarray:
    """Stand-in for `librosa.load(path, sr=sr)` (I_ea/predict.py:79-80)."""
    x, sr_in = read_wav(path)
    return resample(x, sr_in, sr)


def to_int16_pcm(audio: torch.Tensor) -> np.ndarray:
    """`audio * 32768` then `.astype('int16')`: truncation toward zero, as the script does (I_ea/predict.py:204-206).
    Values are clamped to the int16 range first so a full-scale +1.0 cannot wrap."""
    a = (audio.detach().float().cpu() * MAX_WAV_VALUE).clamp(-32768.0, 32767.0)
    return a.numpy().astype("int16")


def write_wav(path: str, pcm_or_float: np.ndarray, sr: int):
    from scipy.io import wavfile
    wavfile.write(path, sr, pcm_or_float)
