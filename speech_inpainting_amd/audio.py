"""Host-side audio glue around the hot path: wav I/O, resampling, peak normalisation and the log-mel front-end.

These are the SURVEY.md section 8(f) "next" rows f-1 / f-3: they surround the three replaced subsystems on the real
predict path and run on the host (torch CPU or GPU tensors), NOT in hand-written HIP yet.  The reference uses
librosa for the resampler, the peak normalisation and the mel filterbank (absent offline); the formulas are restated:

* mel filterbank = `librosa.filters.mel(sr=22050, n_fft=1024, n_mels=80, fmin=0, fmax=8000)` (htk=False, norm='slaney'),
  as called at I_ea/dataset/mel_dump.py:66;
* mel spectrogram = I_ea/dataset/mel_dump.py:40-93: reflect-pad 312, STFT(n_fft 1024, hop 441, Hann, center=False),
  sqrt(re^2 + im^2 + 1e-9), mel basis, log(clamp(., 1e-5));
* resampling: `librosa.load(sr=...)` uses a Kaiser-windowed sinc resampler; here scipy's polyphase resampler with a
  Kaiser window.  Not bit-identical to librosa; documented as such (no in-container oracle for it).
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np
import torch

N_FFT, NUM_MELS, HOP, WIN, PAD, SR22, FMIN, FMAX = 1024, 80, 441, 1024, 312, 22050, 0, 8000   # mel_dump.py:11-20
MAX_WAV_VALUE = 32768.0


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


@lru_cache(maxsize=4)
def mel_filterbank(sr: int = SR22, n_fft: int = N_FFT, n_mels: int = NUM_MELS, fmin: float = FMIN, fmax: float = FMAX) -> np.ndarray:
    """Slaney-style triangular filterbank, area-normalised: (n_mels, 1 + n_fft//2) float32."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(y: torch.Tensor) -> torch.Tensor:
    """get_mel (I_ea/dataset/mel_dump.py:96-98): y (B, n) in [-1, 1] at 22.05 kHz -> (B, 80, Tm) log-mel."""
    basis = torch.from_numpy(mel_filterbank()).to(y.device)
    window = torch.hann_window(WIN, device=y.device)
    y = torch.nn.functional.pad(y.unsqueeze(1), (PAD, PAD), mode="reflect").squeeze(1)
    spec = torch.stft(y, N_FFT, hop_length=HOP, win_length=WIN, window=window, center=False, normalized=False,
                      onesided=True, return_complex=True)
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(basis, mag), min=1e-5))


def peak_normalize(x: np.ndarray, scale: float = 0.95) -> np.ndarray:
    """`librosa.util.normalize(x) * 0.95` (I_ea/predict.py:93,104): divide by max |x| (unchanged if all zero)."""
    m = float(np.max(np.abs(x))) if x.size else 0.0
    return (x / m * scale).astype(np.float32) if m > np.finfo(np.float32).tiny else x.astype(np.float32)


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
