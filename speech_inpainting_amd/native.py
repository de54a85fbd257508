"""ctypes binding of libsi_hip.so (include/si_hip.h).

PyTorch is only plumbing here: it owns device memory (inputs, outputs, workspace) and the HIP stream.
There is no CPU fallback: if the shared library is absent, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from .arch import HubertArch, VocoderArch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SI_HIP_LIB") or os.path.join(_HERE, "libsi_hip.so")   # SI_HIP_LIB: A/B builds of the library

SI_MATH = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2, "fp16": 3, "f16": 3}
SI_MAX_CONV, SI_MAX_UPS, SI_MAX_RB, SI_MAX_DIL = 8, 8, 4, 4

EXPORTS = ["si_version", "si_create", "si_destroy", "si_last_error", "si_load_weights", "si_alloc_weights",
           "si_weights_device_ptr", "si_weights_check", "si_workspace_bytes", "si_hubert_forward", "si_hubert_forward_padded", "si_hubert_forward_varlen", "si_hubert_extract_features", "si_code_splice", "si_codebook_splice",
           "si_codebook_splice_varlen", "si_hifigan_forward_varlen", "si_mel_frontend_varlen",
           "si_codebook_splice_labels", "si_codebook_metrics", "si_kmeans_assign", "si_mel_metrics", "si_sisdr", "si_unit_frontend",
           "si_f0_encoder_weight_floats", "si_f0_encoder_frames", "si_f0_encoder_workspace_bytes", "si_f0_encoder_forward",
           "si_resample_poly", "si_resample_sinc", "si_pcm16", "si_extend_mel", "si_hifigan_forward", "si_mel_frames", "si_mel_workspace_bytes", "si_mel_frontend", "si_num_frames",
           "si_vocoder_samples", "si_profile_start", "si_profile_filter", "si_profile_stop",
           "si_debug_capture", "si_debug_size"]


class _F0EncStruct(C.Structure):
    """Mirror of si_f0enc_desc."""
    _fields_ = [(n, C.c_int32) for n in ("in_width", "out_width", "width", "n_state", "depth", "down_t", "stride_t", "dilation_growth")]


class F0EncDesc:
    """`f0_encoder_params` of I_da/configs/LJSpeech/hubert_lut.json:42-52 (one level): n_state = int(m_conv * width)."""

    def __init__(self, in_width=1, out_width=128, width=32, depth=4, down_t=4, stride_t=2, dilation_growth=3, m_conv=1.0):
        self.in_width, self.out_width, self.width, self.depth = int(in_width), int(out_width), int(width), int(depth)
        self.down_t, self.stride_t, self.dilation_growth = int(down_t), int(stride_t), int(dilation_growth)
        self.n_state = int(m_conv * width)

    def as_struct(self) -> _F0EncStruct:
        return _F0EncStruct(self.in_width, self.out_width, self.width, self.n_state, self.depth, self.down_t, self.stride_t, self.dilation_growth)

    def down_kernel(self):
        s = self.stride_t
        return (2 * s, s // 2) if s % 2 == 0 else (2 * s + 1, s // 2 + 1)          # jukebox.py:54-57


class SincFilter(C.Structure):
    """Mirror of si_sinc_filter."""
    _fields_ = [("struct_size", C.c_int32), ("nwin", C.c_int32), ("num_table", C.c_int32), ("step", C.c_int32), ("n_time", C.c_int32),
                ("reserved", C.c_int32), ("scale", C.c_double), ("ratio", C.c_double), ("win", C.c_void_p), ("dwin", C.c_void_p),
                ("time_reg", C.c_void_p)]


class ExtractDesc(C.Structure):
    """Mirror of si_extract_desc."""
    _fields_ = [("struct_size", C.c_int32), ("output_layer", C.c_int32), ("normalize", C.c_int32), ("reserved", C.c_int32)]


NORMALIZE_MODES = {None: 0, False: 0, "none": 0, "processor": 1, True: 1, "layer_norm": 2}


class ProfileEntry(C.Structure):
    """Mirror of si_profile_entry."""
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("reserved", C.c_int32),
                ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class ModelDesc(C.Structure):
    """Mirror of si_model_desc (include/si_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32),
        ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("intermediate_size", C.c_int32),
        ("num_conv", C.c_int32),
        ("conv_dim", C.c_int32 * SI_MAX_CONV), ("conv_kernel", C.c_int32 * SI_MAX_CONV), ("conv_stride", C.c_int32 * SI_MAX_CONV),
        ("conv_bias", C.c_int32), ("feat_norm_layer", C.c_int32), ("stable_layer_norm", C.c_int32),
        ("pos_conv_kernel", C.c_int32), ("pos_conv_groups", C.c_int32), ("feat_proj_layer_norm", C.c_int32),
        ("layer_norm_eps", C.c_float), ("codebook_dim", C.c_int32), ("num_clusters", C.c_int32),
        ("num_mels", C.c_int32), ("num_ups", C.c_int32),
        ("up_rates", C.c_int32 * SI_MAX_UPS), ("up_kernels", C.c_int32 * SI_MAX_UPS), ("up_initial_channel", C.c_int32),
        ("num_rb", C.c_int32), ("rb_kernels", C.c_int32 * SI_MAX_RB), ("num_dil", C.c_int32),
        ("rb_dilations", (C.c_int32 * SI_MAX_DIL) * SI_MAX_RB),
        ("encoder_math", C.c_int32), ("vocoder_math", C.c_int32), ("vocoder_chunk", C.c_int32), ("resblock_type", C.c_int32),
    ]


def make_desc(harch: HubertArch, varch: VocoderArch, num_clusters: int, encoder_math: str = "fp32",
              vocoder_math: str = "fp32", vocoder_chunk: int = 0) -> ModelDesc:
    d = ModelDesc()
    d.struct_size = C.sizeof(ModelDesc)
    d.hidden_size, d.num_layers = harch.hidden_size, harch.num_hidden_layers
    d.num_heads, d.intermediate_size = harch.num_attention_heads, harch.intermediate_size
    n = len(harch.conv_dim)
    if n > SI_MAX_CONV or not (len(harch.conv_kernel) == len(harch.conv_stride) == n):
        raise ValueError("unsupported conv stack")
    d.num_conv = n
    for i in range(n):
        d.conv_dim[i], d.conv_kernel[i], d.conv_stride[i] = harch.conv_dim[i], harch.conv_kernel[i], harch.conv_stride[i]
    d.conv_bias = int(harch.conv_bias)
    if harch.feat_extract_norm not in ("group", "layer"):
        raise ValueError(f"feat_extract_norm={harch.feat_extract_norm!r}")
    d.feat_norm_layer = int(harch.feat_extract_norm == "layer")
    d.stable_layer_norm = int(harch.do_stable_layer_norm)
    d.pos_conv_kernel, d.pos_conv_groups = harch.num_conv_pos_embeddings, harch.num_conv_pos_embedding_groups
    d.feat_proj_layer_norm = int(harch.feat_proj_layer_norm)
    d.layer_norm_eps = harch.layer_norm_eps
    d.codebook_dim, d.num_clusters = harch.codebook_dim, int(num_clusters)
    d.num_mels = varch.num_mels
    nu = len(varch.upsample_rates)
    if nu > SI_MAX_UPS or len(varch.upsample_kernel_sizes) != nu:
        raise ValueError("unsupported upsample stack")
    d.num_ups = nu
    for i in range(nu):
        d.up_rates[i], d.up_kernels[i] = varch.upsample_rates[i], varch.upsample_kernel_sizes[i]
    d.up_initial_channel = varch.upsample_initial_channel
    nr = len(varch.resblock_kernel_sizes)
    nd = len(varch.resblock_dilation_sizes[0])
    if nr > SI_MAX_RB or nd > SI_MAX_DIL or any(len(x) != nd for x in varch.resblock_dilation_sizes):
        raise ValueError("unsupported resblock shape")
    d.num_rb, d.num_dil = nr, nd
    for j in range(nr):
        d.rb_kernels[j] = varch.resblock_kernel_sizes[j]
        for k in range(nd):
            d.rb_dilations[j][k] = varch.resblock_dilation_sizes[j][k]
    d.encoder_math, d.vocoder_math = SI_MATH[encoder_math], SI_MATH[vocoder_math]
    d.vocoder_chunk = int(vocoder_chunk)
    if str(varch.resblock) not in ("1", "2"):
        raise ValueError(f"resblock={varch.resblock!r}: '1' or '2' (I_ea/hifi_gan/models.py:89)")
    d.resblock_type = int(varch.resblock)
    return d


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libsi_hip.so and declare the prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: the HIP library has not been built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C speech_inpainting_amd/csrc`).  There is no CPU fallback for the inpainting path.")
    lib = C.CDLL(p)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.si_version.restype = i32
    lib.si_create.argtypes = [C.POINTER(vp), i32, C.POINTER(ModelDesc)]
    lib.si_destroy.argtypes = [vp]
    lib.si_destroy.restype = None
    lib.si_last_error.argtypes = [vp]
    lib.si_last_error.restype = C.c_char_p
    lib.si_load_weights.argtypes = [vp, vp, sz, C.c_char_p]
    lib.si_alloc_weights.argtypes = [vp]
    lib.si_weights_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    lib.si_weights_check.argtypes = [vp]
    lib.si_workspace_bytes.argtypes = [vp, i32, i32, i32, C.POINTER(sz)]
    lib.si_hubert_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_hubert_forward_padded.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_hubert_forward_varlen.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_codebook_splice_varlen.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp]
    lib.si_hifigan_forward_varlen.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_mel_frontend_varlen.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_hubert_extract_features.argtypes = [vp, C.POINTER(ExtractDesc), vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.si_code_splice.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.si_codebook_splice.argtypes = [vp, vp, i32, i32, vp, i32, vp, i32, vp, vp]
    lib.si_codebook_splice_labels.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp]
    lib.si_codebook_metrics.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    lib.si_kmeans_assign.argtypes = [vp, vp, C.c_int64, i32, vp, i32, vp, vp, vp]
    lib.si_mel_metrics.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.si_sisdr.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.si_unit_frontend.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, vp, vp]
    lib.si_f0_encoder_weight_floats.argtypes = [vp]
    lib.si_f0_encoder_weight_floats.restype = C.c_size_t
    lib.si_f0_encoder_frames.argtypes = [vp, i32]
    lib.si_f0_encoder_frames.restype = i32
    lib.si_f0_encoder_workspace_bytes.argtypes = [vp, i32, i32]
    lib.si_f0_encoder_workspace_bytes.restype = C.c_size_t
    lib.si_f0_encoder_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, C.c_size_t, vp]
    lib.si_resample_poly.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.si_resample_sinc.argtypes = [vp, vp, vp, i32, i32, C.POINTER(SincFilter), i32, vp, vp]
    lib.si_pcm16.argtypes = [vp, vp, C.c_int64, vp, vp]
    lib.si_extend_mel.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.si_hifigan_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_mel_frames.argtypes = [i32]
    lib.si_mel_workspace_bytes.argtypes = [vp, i32, i32, C.POINTER(sz)]
    lib.si_mel_frontend.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.si_num_frames.argtypes = [vp, i32]
    lib.si_vocoder_samples.argtypes = [vp, i32, i32]
    lib.si_debug_capture.argtypes = [vp, C.c_char_p, vp, C.c_long]
    lib.si_debug_size.argtypes = [vp, C.c_char_p]
    lib.si_debug_size.restype = C.c_long
    lib.si_profile_start.argtypes = [vp, i32]
    lib.si_profile_filter.argtypes = [vp, C.c_char_p]
    lib.si_profile_stop.argtypes = [vp, C.POINTER(ProfileEntry), i32, C.POINTER(i32)]
    for name in EXPORTS:
        if name not in ("si_destroy", "si_last_error", "si_debug_size"):
            getattr(lib, name).restype = i32
    if path is None:
        _lib = lib
    return lib


class NativeError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class NativeContext:
    """One si_ctx: a model pair bound to one GPU."""

    def __init__(self, desc: ModelDesc, device: torch.device):
        self.lib = load_library()
        if device.type != "cuda":
            raise RuntimeError(f"the HIP path needs a GPU device, got {device} (no CPU fallback)")
        self.device = device
        self.desc = desc
        self._h = C.c_void_p(0)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        rc = self.lib.si_create(C.byref(self._h), idx, C.byref(desc))
        if rc != 0:
            raise NativeError(f"si_create failed ({rc}): {self.lib.si_last_error(None).decode()}")
        self._ws: Optional[torch.Tensor] = None

    def close(self):
        """Frees the context and its packed weight blob.  Any tensor from `weights_tensor()` is a VIEW of that blob and
        must not be used after this call (the holder is dropped here so a stale view cannot be handed out again)."""
        self._wholder = None
        self._ws = None
        self._ws_mel = self._ws_mel_old = None
        if getattr(self, "_h", None) and self._h.value:
            self.lib.si_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise NativeError(f"{what} failed ({rc}): {self.lib.si_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- weights
    def load_weights(self, blob: np.ndarray, index: str):
        blob = np.ascontiguousarray(blob)
        self._check(self.lib.si_load_weights(self._h, blob.ctypes.data_as(C.c_void_p), blob.nbytes, index.encode()),
                    "si_load_weights")

    def alloc_weights(self):
        self._check(self.lib.si_alloc_weights(self._h), "si_alloc_weights")

    def weights_check(self):
        """Compare the blob's layout fingerprint with this context's plan (after a broadcast); raises NativeError on a mismatch."""
        self._check(self.lib.si_weights_check(self._h), "si_weights_check")

    def weights_ptr(self):
        """(device address, byte count) of the packed blob as si_weights_device_ptr reports it."""
        p, n = C.c_void_p(0), C.c_size_t(0)
        self._check(self.lib.si_weights_device_ptr(self._h, C.byref(p), C.byref(n)), "si_weights_device_ptr")
        return int(p.value), int(n.value)

    def weights_tensor(self) -> torch.Tensor:
        """The packed device blob as a uint8 tensor VIEW (for the RCCL broadcast): it aliases library-owned memory and
        is valid only while this context is open."""
        p, n = C.c_void_p(0), C.c_size_t(0)
        self._check(self.lib.si_weights_device_ptr(self._h, C.byref(p), C.byref(n)), "si_weights_device_ptr")

        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (n.value,), "typestr": "|u1", "data": (p.value, False), "version": 2}
        self._wholder = h
        t = torch.as_tensor(h, device=self.device)
        if t.data_ptr() != p.value or t.numel() != n.value:      # as_tensor must alias, never copy
            raise NativeError("weights_tensor: torch did not alias the library's weight blob")
        return t

    # ---- shapes / workspace
    def num_frames(self, n: int) -> int:
        return int(self.lib.si_num_frames(self._h, n))

    def mel_frames(self, n22: int) -> int:
        return int(self.lib.si_mel_frames(int(n22)))

    def vocoder_samples(self, tm: int, stretch: bool = True) -> int:
        return int(self.lib.si_vocoder_samples(self._h, tm, int(stretch)))

    def workspace(self, B: int, N: int, Tm: int) -> torch.Tensor:
        need = C.c_size_t(0)
        self._check(self.lib.si_workspace_bytes(self._h, B, N, Tm, C.byref(need)), "si_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- forward calls (all enqueue on torch's current stream)
    def hubert_forward(self, wav: torch.Tensor, mask_start: Optional[torch.Tensor], mask_len: Optional[torch.Tensor],
                       normalize: bool = True, valid_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """valid_len (B,) int32 on the device: real samples of each RIGHT-PADDED clip (None: every clip fills its row)."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        B, N = wav.shape
        T = self.num_frames(N)
        if T < 1:
            raise ValueError(f"clip of {N} samples is too short")
        for m in (mask_start, mask_len, valid_len):
            assert m is None or (m.is_cuda and m.dtype == torch.int32 and m.numel() == B and m.is_contiguous())
        out = torch.empty(B, T, self.desc.codebook_dim, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, N, 0)
        self._check(self.lib.si_hubert_forward_padded(self._h, _ptr(wav), _ptr(mask_start), _ptr(mask_len), _ptr(valid_len),
                                                      int(normalize), B, N, _ptr(out), _ptr(ws), ws.numel(), self._stream()),
                    "si_hubert_forward_padded")
        return out

    @staticmethod
    def _host_lens(lens, B: int):
        """HOST int32 (B) lengths of a ragged batch as a ctypes array (the library reads it during the call only)."""
        a = np.ascontiguousarray(np.asarray(lens, dtype=np.int32).reshape(-1))
        if a.size != B:
            raise ValueError(f"ragged batch: {a.size} lengths for {B} clips")
        return a

    def hubert_forward_varlen(self, wav: torch.Tensor, sample_len, mask_start: Optional[torch.Tensor] = None,
                              mask_len: Optional[torch.Tensor] = None, normalize: bool = True) -> torch.Tensor:
        """RAGGED batch: wav (B, Nmax), clip b = the first sample_len[b] samples of its row (host ints) -> (B, Tmax, D) with zero
        rows past each clip's own frames; every clip's rows equal that clip run alone (si_hubert_forward_varlen)."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        B, N = wav.shape
        lens = self._host_lens(sample_len, B)
        T = self.num_frames(N)
        if T < 1:
            raise ValueError(f"clip of {N} samples is too short")
        for m in (mask_start, mask_len):
            assert m is None or (m.is_cuda and m.dtype == torch.int32 and m.numel() == B and m.is_contiguous())
        out = torch.empty(B, T, self.desc.codebook_dim, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, N, 0)
        self._check(self.lib.si_hubert_forward_varlen(self._h, _ptr(wav), _ptr(mask_start), _ptr(mask_len), lens.ctypes.data_as(C.c_void_p),
                                                      int(normalize), B, N, _ptr(out), _ptr(ws), ws.numel(), self._stream()),
                    "si_hubert_forward_varlen")
        return out

    def codebook_splice_varlen(self, feats: torch.Tensor, frame_pos: torch.Tensor, frame_cnt: torch.Tensor, lm: int, mel: torch.Tensor) -> torch.Tensor:
        """As codebook_splice with a per-clip frame count (B,) int32 on the device; labels past a clip's count are -1."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.dim() == 3
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3
        B, T, D = feats.shape
        for m in (frame_pos, frame_cnt):
            assert m.is_cuda and m.dtype == torch.int32 and m.is_contiguous() and m.numel() == B
        assert mel.shape[0] == B and mel.shape[1] == D
        labels = torch.empty(B, lm, dtype=torch.int64, device=self.device)
        self._check(self.lib.si_codebook_splice_varlen(self._h, _ptr(feats), B, T, _ptr(frame_pos), _ptr(frame_cnt), lm, _ptr(mel), mel.shape[2],
                                                       _ptr(labels), self._stream()), "si_codebook_splice_varlen")
        return labels

    def hifigan_forward_varlen(self, mel: torch.Tensor, mel_len, stretch: bool = True) -> torch.Tensor:
        """RAGGED batch: mel (B, D, Tm_max), clip b = its first mel_len[b] frames (host ints) -> (B, Lmax), the first
        vocoder_samples(mel_len[b]) samples of row b equal that clip's waveform alone, the rest is zero."""
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3
        B, D, Tm = mel.shape
        assert D == self.desc.num_mels
        lens = self._host_lens(mel_len, B)
        L = self.vocoder_samples(Tm, stretch)
        out = torch.empty(B, L, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, 0, Tm)
        self._check(self.lib.si_hifigan_forward_varlen(self._h, _ptr(mel), lens.ctypes.data_as(C.c_void_p), B, Tm, int(stretch), _ptr(out),
                                                       _ptr(ws), ws.numel(), self._stream()), "si_hifigan_forward_varlen")
        return out

    def mel_frontend_varlen(self, wave22: torch.Tensor, sample_len, mask_start: Optional[torch.Tensor] = None,
                            mask_end: Optional[torch.Tensor] = None, normalize: bool = True) -> torch.Tensor:
        """RAGGED batch: wave22 (B, N22max), clip b = its first sample_len[b] samples -> (B, 80, Tm_max), zero frames past a clip's own."""
        assert wave22.is_cuda and wave22.dtype == torch.float32 and wave22.dim() == 2 and wave22.is_contiguous()
        B, N = wave22.shape
        lens = self._host_lens(sample_len, B)
        Tm = int(self.lib.si_mel_frames(N))
        if Tm < 1:
            raise ValueError(f"clip of {N} samples is too short for the mel front-end")
        assert (mask_start is None) == (mask_end is None)
        for m in (mask_start, mask_end):
            assert m is None or (m.is_cuda and m.dtype == torch.int32 and m.numel() == B and m.is_contiguous())
        ws = self._mel_workspace(B, N)
        out = torch.empty(B, 80, Tm, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_mel_frontend_varlen(self._h, _ptr(wave22), _ptr(mask_start), _ptr(mask_end), lens.ctypes.data_as(C.c_void_p),
                                                    int(normalize), B, N, _ptr(out), _ptr(ws), ws.numel(), self._stream()),
                    "si_mel_frontend_varlen")
        return out

    def hubert_extract_features(self, wav: torch.Tensor, output_layer: int, normalize="layer_norm",
                                mask_start: Optional[torch.Tensor] = None, mask_len: Optional[torch.Tensor] = None,
                                pre_mask_add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """wav (B, N) fp32 -> (B, T, H) fp32: the hidden state after `output_layer` transformer layers (fairseq
        `extract_features(output_layer=...)`, I_da/src/hubert_feature_reader.py:60-65).  normalize: "layer_norm" (I_da:
        F.layer_norm over the clip, eps 1e-5), "processor" (I_ea: eps 1e-7) or None.  mask_start / mask_len int32 (B) and
        pre_mask_add float64 (B): `(y + add) * mask` in front of it (I_da/scripts/inpainting.py:186-192)."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        B, N = wav.shape
        T = self.num_frames(N)
        if T < 1:
            raise ValueError(f"clip of {N} samples is too short")
        for m in (mask_start, mask_len):
            assert m is None or (m.is_cuda and m.dtype == torch.int32 and m.numel() == B and m.is_contiguous())
        assert pre_mask_add is None or (pre_mask_add.is_cuda and pre_mask_add.dtype == torch.float64 and pre_mask_add.numel() == B
                                        and pre_mask_add.is_contiguous())
        if normalize not in NORMALIZE_MODES:
            raise ValueError(f"normalize={normalize!r}: expected one of {list(NORMALIZE_MODES)}")
        x = ExtractDesc(C.sizeof(ExtractDesc), int(output_layer), NORMALIZE_MODES[normalize], 0)
        out = torch.empty(B, T, self.desc.hidden_size, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, N, 0)
        self._check(self.lib.si_hubert_extract_features(self._h, C.byref(x), _ptr(wav), _ptr(mask_start), _ptr(mask_len),
                                                        _ptr(pre_mask_add), B, N, _ptr(out), _ptr(ws), ws.numel(), self._stream()),
                    "si_hubert_extract_features")
        return out

    def code_splice(self, code_clean: torch.Tensor, code_masked: torch.Tensor, first: torch.Tensor, last: torch.Tensor) -> torch.Tensor:
        """(B, T) int64 unit series of the clean and the corrupted clip -> the corrupted clip's units inside frames
        [first[b], last[b]), the clean clip's elsewhere (I_da/scripts/inpainting.py:209-214)."""
        for c in (code_clean, code_masked):
            assert c.is_cuda and c.dtype == torch.int64 and c.dim() == 2 and c.is_contiguous()
        assert code_clean.shape == code_masked.shape
        B, T = code_clean.shape
        for m in (first, last):
            assert m.is_cuda and m.dtype == torch.int32 and m.numel() == B and m.is_contiguous()
        out = torch.empty_like(code_masked)
        self._check(self.lib.si_code_splice(self._h, _ptr(code_clean), _ptr(code_masked), _ptr(first), _ptr(last), B, T, _ptr(out),
                                            self._stream()), "si_code_splice")
        return out

    def codebook_splice(self, feats: torch.Tensor, frame_pos: torch.Tensor, lm: int, mel: torch.Tensor) -> torch.Tensor:
        """In-place splice into `mel` (B, D, Tm); returns labels (B, Lm) int64."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.dim() == 3
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3
        assert frame_pos.is_cuda and frame_pos.dtype == torch.int32 and frame_pos.is_contiguous()
        B, T, D = feats.shape
        assert mel.shape[0] == B and mel.shape[1] == D and frame_pos.numel() == B
        labels = torch.empty(B, lm, dtype=torch.int64, device=self.device)
        self._check(self.lib.si_codebook_splice(self._h, _ptr(feats), B, T, _ptr(frame_pos), lm, _ptr(mel), mel.shape[2],
                                                _ptr(labels), self._stream()), "si_codebook_splice")
        return labels

    def codebook_splice_labels(self, labels: torch.Tensor, frame_pos: torch.Tensor, mel: torch.Tensor) -> None:
        """In-place splice of the raw centroids of GIVEN labels (B, Lm) int64 into `mel` (B, D, Tm)."""
        assert labels.is_cuda and labels.dtype == torch.int64 and labels.is_contiguous() and labels.dim() == 2
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3
        assert frame_pos.is_cuda and frame_pos.dtype == torch.int32 and frame_pos.is_contiguous()
        B, lm = labels.shape
        assert mel.shape[0] == B and frame_pos.numel() == B
        self._check(self.lib.si_codebook_splice_labels(self._h, _ptr(labels), B, _ptr(frame_pos), lm, _ptr(mel), mel.shape[2],
                                                       self._stream()), "si_codebook_splice_labels")

    def codebook_metrics(self, feats: torch.Tensor, frame_pos: torch.Tensor, lm: int, target: torch.Tensor):
        """-> (loss (1,), loss_terms (B, Lm), pred_labels (B, Lm) int64, cos_pred_target (B, Lm))."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.dim() == 3
        assert frame_pos.is_cuda and frame_pos.dtype == torch.int32 and frame_pos.is_contiguous()
        B, T, D = feats.shape
        assert target.is_cuda and target.dtype == torch.int64 and target.is_contiguous() and tuple(target.shape) == (B, lm)
        terms = torch.empty(B, lm, dtype=torch.float32, device=self.device)
        cpt = torch.empty(B, lm, dtype=torch.float32, device=self.device)
        pred = torch.empty(B, lm, dtype=torch.int64, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_codebook_metrics(self._h, _ptr(feats), B, T, _ptr(frame_pos), lm, _ptr(target), _ptr(terms),
                                                 _ptr(loss), _ptr(pred), _ptr(cpt), self._stream()), "si_codebook_metrics")
        return loss, terms, pred, cpt

    def kmeans_assign(self, feats: torch.Tensor, centroids: torch.Tensor, with_distance: bool = False):
        """feats (..., D), centroids (K, D) -> int64 labels (...) [, squared distance to the winner]."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous()
        assert centroids.is_cuda and centroids.dtype == torch.float32 and centroids.is_contiguous() and centroids.dim() == 2
        D = feats.shape[-1]
        assert centroids.shape[1] == D
        rows = feats.numel() // D
        labels = torch.empty(feats.shape[:-1], dtype=torch.int64, device=self.device)
        dist = torch.empty(feats.shape[:-1], dtype=torch.float32, device=self.device) if with_distance else None
        self._check(self.lib.si_kmeans_assign(self._h, _ptr(feats), rows, D, _ptr(centroids), centroids.shape[0], _ptr(labels),
                                              _ptr(dist), self._stream()), "si_kmeans_assign")
        return (labels, dist) if with_distance else labels

    def mel_metrics(self, mel_a: torch.Tensor, mel_b: torch.Tensor, center: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mel_a, mel_b (B, D, L) -> (B, 3) = avg_cosine_sim, avg_d2_dist, rmse of I_ea/metrics.py:38-62 per clip."""
        for m in (mel_a, mel_b):
            assert m.is_cuda and m.dtype == torch.float32 and m.dim() == 3 and m.is_contiguous()
        assert mel_a.shape == mel_b.shape
        B, D, L = mel_a.shape
        assert center is None or (center.is_cuda and center.dtype == torch.float32 and center.numel() == D and center.is_contiguous())
        out = torch.empty(B, 3, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_mel_metrics(self._h, _ptr(mel_a), _ptr(mel_b), B, D, L, _ptr(center), _ptr(out), self._stream()), "si_mel_metrics")
        return out

    def sisdr(self, est: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
        """est, ref (B, n) waveforms -> (B,) SI-SDR in dB (I_ea/metrics.py:127-142)."""
        for m in (est, ref):
            assert m.is_cuda and m.dtype == torch.float32 and m.dim() == 2 and m.is_contiguous()
        assert est.shape == ref.shape
        out = torch.empty(est.shape[0], dtype=torch.float32, device=self.device)
        self._check(self.lib.si_sisdr(self._h, _ptr(est), _ptr(ref), est.shape[0], est.shape[1], _ptr(out), self._stream()), "si_sisdr")
        return out

    def unit_frontend(self, code: torch.Tensor, emb_c: torch.Tensor, f0_code: Optional[torch.Tensor] = None,
                      emb_p: Optional[torch.Tensor] = None, spk_emb: Optional[torch.Tensor] = None) -> torch.Tensor:
        """code (B, Fc) int64, emb_c (Kc, E); optional f0_code (B, Fp) int64 + emb_p (Kp, E); optional spk_emb (B, E)
        -> (B, nparts * E, max(Fc, Fp)) fp32, the CodeGenerator's generator input."""
        assert code.is_cuda and code.dtype == torch.int64 and code.dim() == 2 and code.is_contiguous()
        assert emb_c.is_cuda and emb_c.dtype == torch.float32 and emb_c.dim() == 2 and emb_c.is_contiguous()
        B, Fc = code.shape
        Kc, E = emb_c.shape
        Fp, Kp = 0, 0
        if f0_code is not None:
            assert f0_code.is_cuda and f0_code.dtype == torch.int64 and f0_code.is_contiguous() and f0_code.shape[0] == B
            assert emb_p is not None and emb_p.is_cuda and emb_p.dtype == torch.float32 and emb_p.is_contiguous() and emb_p.shape[1] == E
            Fp, Kp = f0_code.shape[1], emb_p.shape[0]
        if spk_emb is not None:
            assert spk_emb.is_cuda and spk_emb.dtype == torch.float32 and spk_emb.is_contiguous() and tuple(spk_emb.shape) == (B, E)
        nparts = 1 + (f0_code is not None) + (spk_emb is not None)
        out = torch.empty(B, nparts * E, max(Fc, Fp), dtype=torch.float32, device=self.device)
        self._check(self.lib.si_unit_frontend(self._h, _ptr(code), Fc, _ptr(f0_code), Fp, _ptr(spk_emb), _ptr(emb_c), Kc, _ptr(emb_p), Kp,
                                              E, B, _ptr(out), self._stream()), "si_unit_frontend")
        return out

    def f0_encoder(self, desc: "F0EncDesc", weights: torch.Tensor, f0: torch.Tensor) -> torch.Tensor:
        """f0 (B, in_width, T) fp32, weights = the packed encoder parameters (see include/si_hip.h) -> (B, T', out_width)
        fp32 channels-last: the rows the VQ bottleneck quantises (I_da/src/model.py:160-163)."""
        assert f0.is_cuda and f0.dtype == torch.float32 and f0.dim() == 3 and f0.is_contiguous() and f0.shape[1] == desc.in_width
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.dim() == 1 and weights.is_contiguous()
        d = desc.as_struct()
        need = int(self.lib.si_f0_encoder_weight_floats(C.byref(d)))
        if weights.numel() != need:
            raise ValueError(f"F0 encoder weights: {weights.numel()} floats, the descriptor needs {need}")
        B, _, T = f0.shape
        Tp = int(self.lib.si_f0_encoder_frames(C.byref(d), T))
        if Tp <= 0:
            raise ValueError(f"{T} F0 frames are too few for this encoder")
        ws = torch.empty(int(self.lib.si_f0_encoder_workspace_bytes(C.byref(d), B, T)), dtype=torch.uint8, device=self.device)
        out = torch.empty(B, Tp, desc.out_width, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_f0_encoder_forward(self._h, C.byref(d), _ptr(weights), _ptr(f0), B, T, _ptr(out), _ptr(ws), ws.numel(),
                                                   self._stream()), "si_f0_encoder_forward")
        return out

    def resample_poly(self, x: torch.Tensor, taps: torch.Tensor, up: int, down: int, pre_remove: int, n_out: int) -> torch.Tensor:
        """x (B, n_in) -> (B, n_out): upfirdn(taps, x, up, down)[pre_remove : pre_remove + n_out] (see audio.design_resampler)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
        assert taps.is_cuda and taps.dtype == torch.float32 and taps.dim() == 1 and taps.is_contiguous()
        y = torch.empty(x.shape[0], n_out, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_resample_poly(self._h, _ptr(x), x.shape[0], x.shape[1], _ptr(taps), taps.numel(), int(up), int(down),
                                              int(pre_remove), int(n_out), _ptr(y), self._stream()), "si_resample_poly")
        return y

    def resample_sinc(self, x: torch.Tensor, filt: dict, n_out: int, n_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x (B, n_in) -> (B, n_out) by resampy's `kaiser_best` interpolation (librosa 0.9.1's resampler); `filt` holds the DEVICE
        tables of audio.design_kaiser_best (win, dwin, time_reg float64) and its scalars; n_len (B,) int32 device lengths or None."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
        for k in ("win", "dwin", "time_reg"):
            assert filt[k].is_cuda and filt[k].dtype == torch.float64 and filt[k].is_contiguous()
        assert n_len is None or (n_len.is_cuda and n_len.dtype == torch.int32 and n_len.numel() == x.shape[0] and n_len.is_contiguous())
        f = SincFilter(C.sizeof(SincFilter), filt["win"].numel(), int(filt["num_table"]), int(filt["step"]), filt["time_reg"].numel(), 0,
                       float(filt["scale"]), float(filt["ratio"]), filt["win"].data_ptr(), filt["dwin"].data_ptr(), filt["time_reg"].data_ptr())
        y = torch.empty(x.shape[0], int(n_out), dtype=torch.float32, device=self.device)
        self._check(self.lib.si_resample_sinc(self._h, _ptr(x), _ptr(n_len), x.shape[0], x.shape[1], C.byref(f), int(n_out), _ptr(y), self._stream()),
                    "si_resample_sinc")
        return y

    def pcm16(self, wav: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 waveform (any shape) -> int16 PCM on the device: `audio * 32768` truncated toward zero (I_ea/predict.py:204-206)."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.is_contiguous()
        if out is None:
            out = torch.empty(wav.shape, dtype=torch.int16, device=self.device)
        assert out.is_cuda and out.dtype == torch.int16 and out.is_contiguous() and out.numel() == wav.numel()
        self._check(self.lib.si_pcm16(self._h, _ptr(wav), wav.numel(), _ptr(out), self._stream()), "si_pcm16")
        return out

    def extend_mel(self, mel: torch.Tensor) -> torch.Tensor:
        """(B, D, Tm) -> (B, D, floor(Tm * 441 / 256)): the x441/256 stretch alone (the generator's input with stretch=False)."""
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3 and mel.shape[1] == self.desc.num_mels
        B, D, Tm = mel.shape
        hop = 1
        for i in range(self.desc.num_ups):
            hop *= self.desc.up_rates[i]
        out = torch.empty(B, D, self.vocoder_samples(Tm, True) // hop, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_extend_mel(self._h, _ptr(mel), B, Tm, _ptr(out), self._stream()), "si_extend_mel")
        return out

    def hifigan_forward(self, mel: torch.Tensor, stretch: bool = True) -> torch.Tensor:
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3
        B, D, Tm = mel.shape
        assert D == self.desc.num_mels
        L = self.vocoder_samples(Tm, stretch)
        out = torch.empty(B, L, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, 0, Tm)
        self._check(self.lib.si_hifigan_forward(self._h, _ptr(mel), B, Tm, int(stretch), _ptr(out), _ptr(ws), ws.numel(),
                                                self._stream()), "si_hifigan_forward")
        return out

    def _mel_workspace(self, B: int, N: int) -> torch.Tensor:
        """The mel front-end's OWN scratch (not the encoder / vocoder workspace: the front-end may run on a side stream under the
        encoder).  Grown only; a buffer that is replaced stays referenced until the next call, so work still queued on it is safe."""
        need = C.c_size_t(0)
        self._check(self.lib.si_mel_workspace_bytes(self._h, B, N, C.byref(need)), "si_mel_workspace_bytes")
        cur = getattr(self, "_ws_mel", None)
        if cur is None or cur.numel() < need.value:
            self._ws_mel_old = cur
            self._ws_mel = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws_mel

    def mel_frontend(self, wave22: torch.Tensor, mask_start: Optional[torch.Tensor] = None,
                     mask_end: Optional[torch.Tensor] = None, normalize: bool = True) -> torch.Tensor:
        """(B, N22) raw 22.05 kHz clips -> (B, 80, Tm) log-mel; the span [mask_start, mask_end) of each clip is zeroed first."""
        assert wave22.is_cuda and wave22.dtype == torch.float32 and wave22.dim() == 2 and wave22.is_contiguous()
        B, N = wave22.shape
        Tm = int(self.lib.si_mel_frames(N))
        if Tm < 1:
            raise ValueError(f"clip of {N} samples is too short for the mel front-end")
        assert (mask_start is None) == (mask_end is None)
        for m in (mask_start, mask_end):
            assert m is None or (m.is_cuda and m.dtype == torch.int32 and m.numel() == B and m.is_contiguous())
        ws = self._mel_workspace(B, N)
        out = torch.empty(B, 80, Tm, dtype=torch.float32, device=self.device)
        self._check(self.lib.si_mel_frontend(self._h, _ptr(wave22), _ptr(mask_start), _ptr(mask_end), int(normalize), B, N,
                                             _ptr(out), _ptr(ws), ws.numel(), self._stream()), "si_mel_frontend")
        return out

    def capture(self, names, capacity: int = 0):
        """Register capture buffers for the named intermediates (sizes come from the previous forward unless
        `capacity` floats is given).  Returns {name: tensor}; the tensors are filled by the next forward."""
        out = {}
        for nm in names:
            n = capacity or self.lib.si_debug_size(self._h, nm.encode())
            if n < 0:
                raise NativeError(self.lib.si_last_error(self._h).decode())
            t = torch.zeros(n, dtype=torch.float32, device=self.device)
            self._check(self.lib.si_debug_capture(self._h, nm.encode(), _ptr(t), n), "si_debug_capture")
            out[nm] = t
        self._captures = out
        return out

    def clear_captures(self):
        for nm in list(getattr(self, "_captures", {})):
            self.lib.si_debug_capture(self._h, nm.encode(), C.c_void_p(0), 0)
        self._captures = {}

    # ---- per-kernel HIP-event timing
    def profile_start(self, max_launches: int = 20000):
        self._check(self.lib.si_profile_start(self._h, int(max_launches)), "si_profile_start")

    def profile_filter(self, family: Optional[str]):
        """Bracket only launches of this kernel family (None: all)."""
        self._check(self.lib.si_profile_filter(self._h, family.encode() if family else None), "si_profile_filter")

    def profile_stop(self):
        """-> list of dicts {name, launches, ms, flops, bytes}; waits for the recorded events."""
        cap = 128
        arr = (ProfileEntry * cap)()
        n = C.c_int(0)
        self._check(self.lib.si_profile_stop(self._h, arr, cap, C.byref(n)), "si_profile_stop")
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops,
                     bytes=arr[i].bytes) for i in range(min(n.value, cap))]
