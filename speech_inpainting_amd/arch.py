"""Architecture descriptors for the I_ea hot path.

These are plain dataclasses holding the dimensions the native library needs.  They are
filled either from a HuggingFace HuBERT ``config.json`` (the schema dumped in the
reference at I_ea/dataset/config.json:62-124) and a HiFi-GAN ``config.json``
(I_ea/hifi_gan/config_v1.json:1-37), or from the built-in presets below, which restate
the two checkpoints the reference names (I_ea/model.py:26-31).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, asdict
from typing import List, Tuple


@dataclass
class HubertArch:
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"       # "group" (base) | "layer" (large)
    do_stable_layer_norm: bool = False     # False: post-LN (base); True: pre-LN (large)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    feat_proj_layer_norm: bool = True
    codebook_dim: int = 80                 # final_layers Linear(H -> codebook_dim), I_ea/model.py:75-78

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def feat_lengths(self, n: int) -> List[int]:
        """Conv-stack lengths [N, L1, ..., T] (formula: modeling_hubert.py:664-677)."""
        out = [n]
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
            out.append(n)
        return out

    def num_frames(self, n: int) -> int:
        return self.feat_lengths(n)[-1]

    @classmethod
    def base(cls) -> "HubertArch":
        """facebook/hubert-base-ls960 (I_ea/model.py:28)."""
        return cls()

    @classmethod
    def large(cls) -> "HubertArch":
        """facebook/hubert-large-ls960-ft (I_ea/model.py:31)."""
        return cls(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                   intermediate_size=4096, conv_bias=True, feat_extract_norm="layer",
                   do_stable_layer_norm=True)

    @classmethod
    def tiny(cls, **kw) -> "HubertArch":
        """Small shape used by fast parity tests (same code paths as base)."""
        d = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                 intermediate_size=256, conv_dim=(32,) * 7,
                 num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4,
                 codebook_dim=80)
        d.update(kw)
        return cls(**d)

    @classmethod
    def from_hf_config(cls, cfg: dict, codebook_dim: int = 80) -> "HubertArch":
        if cfg.get("model_type", "hubert") != "hubert":
            raise ValueError(f"not a HuBERT config (model_type={cfg.get('model_type')!r})")
        if cfg.get("feat_extract_activation", "gelu") != "gelu" or cfg.get("hidden_act", "gelu") != "gelu":
            raise ValueError("only the erf-GELU activation of the reference checkpoints is supported")
        if cfg.get("conv_pos_batch_norm", False):
            raise ValueError("conv_pos_batch_norm checkpoints are not supported")
        return cls(
            hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
            num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
            conv_dim=tuple(cfg["conv_dim"]), conv_kernel=tuple(cfg["conv_kernel"]),
            conv_stride=tuple(cfg["conv_stride"]), conv_bias=bool(cfg.get("conv_bias", False)),
            feat_extract_norm=cfg.get("feat_extract_norm", "group"),
            do_stable_layer_norm=bool(cfg.get("do_stable_layer_norm", False)),
            num_conv_pos_embeddings=cfg.get("num_conv_pos_embeddings", 128),
            num_conv_pos_embedding_groups=cfg.get("num_conv_pos_embedding_groups", 16),
            layer_norm_eps=float(cfg.get("layer_norm_eps", 1e-5)),
            feat_proj_layer_norm=bool(cfg.get("feat_proj_layer_norm", True)),
            codebook_dim=codebook_dim)

    def to_hf(self) -> dict:
        """The HuggingFace `config.json` dict of this architecture (the schema of I_ea/dataset/config.json:62-124)."""
        return dict(model_type="hubert", hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
                    num_attention_heads=self.num_attention_heads, intermediate_size=self.intermediate_size,
                    conv_dim=list(self.conv_dim), conv_kernel=list(self.conv_kernel), conv_stride=list(self.conv_stride),
                    conv_bias=self.conv_bias, feat_extract_norm=self.feat_extract_norm, feat_extract_activation="gelu",
                    hidden_act="gelu", do_stable_layer_norm=self.do_stable_layer_norm,
                    num_conv_pos_embeddings=self.num_conv_pos_embeddings,
                    num_conv_pos_embedding_groups=self.num_conv_pos_embedding_groups, layer_norm_eps=self.layer_norm_eps,
                    feat_proj_layer_norm=self.feat_proj_layer_norm, num_feat_extract_layers=len(self.conv_dim))

    @classmethod
    def from_json(cls, path: str, codebook_dim: int = 80) -> "HubertArch":
        with open(path) as f:
            return cls.from_hf_config(json.load(f), codebook_dim)


@dataclass
class VocoderArch:
    """HiFi-GAN generator hyper-parameters (I_ea/hifi_gan/config_v1.json:2,11-15)."""
    resblock: str = "1"
    upsample_rates: Tuple[int, ...] = (8, 8, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 4, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5),) * 3
    num_mels: int = 80
    sampling_rate: int = 22050

    @property
    def hop(self) -> int:
        h = 1
        for u in self.upsample_rates:
            h *= u
        return h

    @classmethod
    def v1(cls) -> "VocoderArch":
        return cls()

    @classmethod
    def tiny(cls) -> "VocoderArch":
        return cls(upsample_initial_channel=256)

    @classmethod
    def v3(cls) -> "VocoderArch":
        """I_ea/hifi_gan/config_v3.json: ResBlock2 blocks (one dilated conv per dilation), three upsamplers."""
        return cls(resblock="2", upsample_rates=(8, 8, 4), upsample_kernel_sizes=(16, 16, 8), upsample_initial_channel=256,
                   resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)))

    @classmethod
    def from_config(cls, h: dict) -> "VocoderArch":
        if str(h.get("resblock", "1")) not in ("1", "2"):
            raise ValueError(f"resblock={h.get('resblock')!r}: the reference knows '1' and '2' (I_ea/hifi_gan/models.py:89)")
        return cls(resblock=str(h.get("resblock", "1")),
                   upsample_rates=tuple(h["upsample_rates"]),
                   upsample_kernel_sizes=tuple(h["upsample_kernel_sizes"]),
                   upsample_initial_channel=int(h["upsample_initial_channel"]),
                   resblock_kernel_sizes=tuple(h["resblock_kernel_sizes"]),
                   resblock_dilation_sizes=tuple(tuple(d) for d in h["resblock_dilation_sizes"]),
                   num_mels=int(h.get("num_mels", 80)),
                   sampling_rate=int(h.get("sampling_rate", 22050)))


EXTEND_NUM, EXTEND_DEN = 441, 256   # extend_mel scale 441/256 (I_ea/hifi_gan/inference_modified.py:17)


MEL_PAD = 312   # padd_ of I_ea/dataset/mel_dump.py:15 (NOT (n_fft - hop) / 2 = 291, which is commented out at :71)


def mel_frames(n22: int, n_fft: int = 1024, hop: int = 441, pad: int = MEL_PAD) -> int:
    """Mel frame count of get_mel (I_ea/dataset/mel_dump.py:72,75-87): reflect-pad 312 each side, STFT center=False."""
    return (n22 + 2 * pad - n_fft) // hop + 1


def extended_frames(tm: int) -> int:
    """Output width of extend_mel: floor(Tm * 441/256) (F.interpolate with scale_factor)."""
    import math
    return int(math.floor(float(tm) * (EXTEND_NUM / EXTEND_DEN)))


def algorithmic_gmac(harch: "HubertArch", varch: "VocoderArch", n16: int, tm: int, head: bool = True, stretch: bool = True):
    """Algorithmic multiply-accumulates (in 1e9) of one clip from layer shapes only -- SURVEY.md 8(d): conv = Cout * Cin * k * Lout,
    convT = Cin * Cout * k * Lin, attention = 4 T H^2 + 2 T^2 H, FFN = 8 T H^2; no padding / halo / recompute.  n16 samples at
    16 kHz, tm mel frames -> (encoder [+ head], vocoder).  4 s / 200 frames, base + V1: (28.47, 105.63) = BASELINE.md section 2."""
    L = harch.feat_lengths(n16)
    T = L[-1]
    macs, cin = 0, 1
    for i, (c, k) in enumerate(zip(harch.conv_dim, harch.conv_kernel)):
        macs += c * cin * k * L[i + 1]
        cin = c
    H, I = harch.hidden_size, harch.intermediate_size
    macs += T * cin * H + T * H * (H // harch.num_conv_pos_embedding_groups) * harch.num_conv_pos_embeddings
    macs += harch.num_hidden_layers * (4 * T * H * H + 2 * T * T * H + 2 * T * H * I)
    if head:
        macs += T * H * harch.codebook_dim
    Lr = extended_frames(tm) if stretch else tm
    c = varch.upsample_initial_channel
    vm = Lr * c * varch.num_mels * 7
    for u, k in zip(varch.upsample_rates, varch.upsample_kernel_sizes):
        vm += c * (c // 2) * k * Lr
        c //= 2
        Lr *= u
        for rk, dils in zip(varch.resblock_kernel_sizes, varch.resblock_dilation_sizes):
            vm += len(dils) * (2 if str(varch.resblock) == "1" else 1) * c * c * rk * Lr
    vm += c * 7 * Lr
    return macs / 1e9, vm / 1e9
