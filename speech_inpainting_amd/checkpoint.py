"""Checkpoint readers for the three artefacts the reference's predict script loads.

* HuBERT: ``torch.load(model_checkpoint)`` of ``CustomModel.state_dict()`` (I_ea/predict.py:149; written by
  I_ea/main.py:264) -> keys ``base_model.<hf key>`` + ``final_layers.*``; or a LOCAL HuggingFace directory
  (``config.json`` + ``model.safetensors`` / ``pytorch_model.bin``) for the encoder weights.  The reference builds
  the architecture from ``HubertConfig.from_pretrained(<hub name>)`` (I_ea/model.py:39) -- a network fetch, which is
  not available here: pass a local path.
* HiFi-GAN: ``torch.load(checkpoint_file)['generator']`` with ``config.json`` beside it (I_ea/predict.py:109-119).
* Codebook: joblib pickle of sklearn ``MiniBatchKMeans`` (``cluster_centers_`` (K, 80), I_ea/dataset/km_label.py:13-14)
  or a plain ``.npy``.

``flatten_checkpoint`` turns them into the (blob, index) pair of ``si_load_weights``; weight-norm folding and
re-layout happen inside the native library, not here.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from .arch import HubertArch, VocoderArch

HUB_NAMES = {"base": "facebook/hubert-base-ls960", "large": "facebook/hubert-large-ls960-ft"}   # I_ea/model.py:26-31


def arch_for_type(model_type: str) -> HubertArch:
    """`hubert_model.type` of predict.yaml -> architecture (anything but 'base' means large, I_ea/model.py:26-31)."""
    return HubertArch.base() if model_type == "base" else HubertArch.large()


def normalize_hubert_keys(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept a CustomModel state dict, or a bare HubertModel / HubertForCTC one (adds the ``base_model.`` prefix)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("masked_spec_embed") or k.startswith("lm_head."):
            continue                                    # unused in eval (modeling_hubert.py:852)
        if k.startswith("base_model.") or k.startswith("final_layers."):
            out[k] = v
        elif k.startswith("hubert."):
            out["base_model." + k[len("hubert."):]] = v
        else:
            out["base_model." + k] = v
    return out


def load_hubert_checkpoint(path: str, model_type: Optional[str] = None) -> Tuple[Dict[str, torch.Tensor], Optional[HubertArch]]:
    """Returns (state dict with CustomModel key names, arch or None when the file carries no config)."""
    if not os.path.exists(path):
        if "/" in path and not path.startswith((".", "/")) and path.count("/") == 1:
            raise FileNotFoundError(
                f"'{path}' looks like a HuggingFace hub name; fetching is not supported offline -- pass a local path "
                "(a CustomModel .pt, or a directory with config.json + model.safetensors)")
        raise FileNotFoundError(path)
    arch = None
    if os.path.isdir(path):
        cfgp = os.path.join(path, "config.json")
        if os.path.exists(cfgp):
            arch = HubertArch.from_json(cfgp)
        st, bn = os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(bn):
            sd = torch.load(bn, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"{path}: neither model.safetensors nor pytorch_model.bin")
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd:
            sd = sd["state_dict"]
    if arch is None and model_type is not None:
        arch = arch_for_type(model_type)
    return normalize_hubert_keys(sd), arch


def fresh_final_layers(arch: HubertArch, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """`final_layers = Sequential(LayerNorm(H), Linear(H, codebook_dim))` as the reference's constructor leaves it when it
    starts from a pretrained encoder (`load_pretrained=True`, I_ea/model.py:26-40,75-78): PyTorch's default initialisation.
    A HuggingFace directory holds the encoder only; the trained head lives in the CustomModel .pt (I_ea/predict.py:149)."""
    g = torch.Generator().manual_seed(seed)
    H, D = arch.hidden_size, arch.codebook_dim
    bound = H ** -0.5                       # kaiming_uniform(a = sqrt(5)) on (D, H) and the bias bound 1 / sqrt(fan_in)
    return {"final_layers.0.weight": torch.ones(H), "final_layers.0.bias": torch.zeros(H),
            "final_layers.1.weight": (torch.rand(D, H, generator=g) * 2 - 1) * bound,
            "final_layers.1.bias": (torch.rand(D, generator=g) * 2 - 1) * bound}


def load_generator_checkpoint(path: str) -> Tuple[Dict[str, torch.Tensor], VocoderArch]:
    """`hifi_gan.checkpoint_file` of predict.yaml; `config.json` must sit beside it (I_ea/predict.py:110-115)."""
    cfg = os.path.join(os.path.split(path)[0], "config.json")
    with open(cfg) as f:
        varch = VocoderArch.from_config(json.load(f))
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if "generator" not in ck:
        raise KeyError(f"{path}: no 'generator' entry (I_ea/predict.py:119)")
    return dict(ck["generator"]), varch


def load_codebook(path: str) -> torch.Tensor:
    """(K, D) fp32 cluster centres from `<km_model_path>/km_model_<K>/model.km` (joblib) or a .npy."""
    if path.endswith(".npy"):
        c = np.load(path)
    else:
        import joblib
        c = joblib.load(path).cluster_centers_
    c = np.asarray(c, dtype=np.float32)
    if c.ndim != 2:
        raise ValueError(f"{path}: centroids must be (K, D), got {c.shape}")
    return torch.from_numpy(np.ascontiguousarray(c))


def flatten_checkpoint(hubert_sd: Mapping[str, torch.Tensor], gen_sd: Mapping[str, torch.Tensor],
                       codebook: torch.Tensor) -> Tuple[np.ndarray, str]:
    """-> (fp32 blob, index text) in the format si_load_weights documents."""
    items = []
    for k, v in normalize_hubert_keys(hubert_sd).items():
        items.append((k, v))
    for k, v in gen_sd.items():
        items.append(("generator." + k, v))
    items.append(("codebook", codebook))
    total = sum(int(v.numel()) for _, v in items)
    blob = np.empty(total, dtype=np.float32)
    lines, off = [], 0
    for name, v in items:
        a = v.detach().to(torch.float32).cpu().contiguous().numpy().reshape(-1)
        blob[off:off + a.size] = a
        lines.append(" ".join([name, str(off * 4), str(v.dim())] + [str(int(s)) for s in v.shape]))
        off += a.size
    return blob, "\n".join(lines) + "\n"
