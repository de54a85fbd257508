"""Deterministic synthetic checkpoints and inputs.

No model weights ship with the reference (they are fetched from HuggingFace / Google Drive,
I_ea/README.md:50), so parity and throughput are established on seeded synthetic weights that
use the reference's exact state-dict key schema:

* HuBERT: ``CustomModel.state_dict()`` = ``base_model.<hf key>`` + ``final_layers.{0,1}.*``
  (I_ea/model.py:40,75-78; saved by I_ea/main.py:264).
* HiFi-GAN: ``{'generator': Generator.state_dict()}`` with weight-norm ``*_g/*_v`` pairs
  (I_ea/hifi_gan/models.py:87-105; loaded at I_ea/predict.py:118-119).
* Codebook: (K, 80) cluster centres (I_ea/dataset/km_label.py:13-14).

Initial scales are chosen so activations stay O(1) through the stack (the reference's default
``init_weights`` std=0.01 gives a near-silent generator, which would make an absolute RMS gate
meaningless).  Everything is drawn from one ``torch.Generator`` in a fixed key order, so the same
seed reproduces the same tensors on any machine with this torch build.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import torch

from .arch import HubertArch, VocoderArch

DEFAULT_SEED = 1234  # the reference's own seed (config_v1.json:9, config.yaml:3)


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def _n(g, *shape, std=1.0, mean=0.0):
    return torch.randn(*shape, generator=g, dtype=torch.float32) * std + mean


def synth_hubert_state(arch: HubertArch, seed: int = DEFAULT_SEED, pos_conv_style: str = "parametrizations"
                       ) -> "OrderedDict[str, torch.Tensor]":
    """CustomModel-style state dict.  pos_conv_style: 'parametrizations' (torch>=2.1 key names) or
    'legacy' (weight_g / weight_v, the names written by the reference's pinned torch 2.0.1)."""
    g = _gen(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    P = "base_model."
    H = arch.hidden_size
    cin = 1
    for i, (c, k) in enumerate(zip(arch.conv_dim, arch.conv_kernel)):
        sd[f"{P}feature_extractor.conv_layers.{i}.conv.weight"] = _n(g, c, cin, k, std=math.sqrt(2.0 / (cin * k)))
        if arch.conv_bias:
            sd[f"{P}feature_extractor.conv_layers.{i}.conv.bias"] = _n(g, c, std=0.1)
        if (arch.feat_extract_norm == "group" and i == 0) or arch.feat_extract_norm == "layer":
            sd[f"{P}feature_extractor.conv_layers.{i}.layer_norm.weight"] = _n(g, c, std=0.1, mean=1.0)
            sd[f"{P}feature_extractor.conv_layers.{i}.layer_norm.bias"] = _n(g, c, std=0.1)
        cin = c
    if arch.feat_proj_layer_norm:
        sd[f"{P}feature_projection.layer_norm.weight"] = _n(g, cin, std=0.1, mean=1.0)
        sd[f"{P}feature_projection.layer_norm.bias"] = _n(g, cin, std=0.1)
    sd[f"{P}feature_projection.projection.weight"] = _n(g, H, cin, std=1.0 / math.sqrt(cin))
    sd[f"{P}feature_projection.projection.bias"] = _n(g, H, std=0.1)
    # positional conv, weight-normed over dim=2 (modeling_hubert.py:78)
    kp, gp = arch.num_conv_pos_embeddings, arch.num_conv_pos_embedding_groups
    v = _n(g, H, H // gp, kp, std=1.0 / math.sqrt((H // gp) * kp))
    gnorm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt() * _n(g, 1, 1, kp, std=0.1, mean=1.0)
    sd[f"{P}encoder.pos_conv_embed.conv.bias"] = _n(g, H, std=0.1)
    if pos_conv_style == "parametrizations":
        sd[f"{P}encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = gnorm
        sd[f"{P}encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = v
    else:
        sd[f"{P}encoder.pos_conv_embed.conv.weight_g"] = gnorm
        sd[f"{P}encoder.pos_conv_embed.conv.weight_v"] = v
    sd[f"{P}encoder.layer_norm.weight"] = _n(g, H, std=0.1, mean=1.0)
    sd[f"{P}encoder.layer_norm.bias"] = _n(g, H, std=0.1)
    I = arch.intermediate_size
    for l in range(arch.num_hidden_layers):
        L = f"{P}encoder.layers.{l}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sd[f"{L}attention.{nm}.weight"] = _n(g, H, H, std=1.0 / math.sqrt(H))
            sd[f"{L}attention.{nm}.bias"] = _n(g, H, std=0.1)
        sd[f"{L}layer_norm.weight"] = _n(g, H, std=0.1, mean=1.0)
        sd[f"{L}layer_norm.bias"] = _n(g, H, std=0.1)
        sd[f"{L}feed_forward.intermediate_dense.weight"] = _n(g, I, H, std=1.0 / math.sqrt(H))
        sd[f"{L}feed_forward.intermediate_dense.bias"] = _n(g, I, std=0.1)
        sd[f"{L}feed_forward.output_dense.weight"] = _n(g, H, I, std=1.0 / math.sqrt(I))
        sd[f"{L}feed_forward.output_dense.bias"] = _n(g, H, std=0.1)
        sd[f"{L}final_layer_norm.weight"] = _n(g, H, std=0.1, mean=1.0)
        sd[f"{L}final_layer_norm.bias"] = _n(g, H, std=0.1)
    sd["final_layers.0.weight"] = _n(g, H, std=0.1, mean=1.0)
    sd["final_layers.0.bias"] = _n(g, H, std=0.1)
    sd["final_layers.1.weight"] = _n(g, arch.codebook_dim, H, std=1.0 / math.sqrt(H))
    sd["final_layers.1.bias"] = _n(g, arch.codebook_dim, std=0.1)
    return sd


def centre_head(sd, vbar: torch.Tensor):
    """A copy of a CustomModel state dict whose head output is shifted by -vbar (`final_layers.1.bias -= vbar`).  With random
    weights every encoder frame maps to nearly the same 80-dim vector, so the cosine arg-max against the centred centroids
    (I_ea/loss_fn.py:44-47) returns one codeword for all frames; centring the head on the frames of interest -- what a head
    trained against the centred centroids emits -- makes the label comparison a comparison of real decisions."""
    out = OrderedDict(sd)
    out["final_layers.1.bias"] = (sd["final_layers.1.bias"].float() - vbar.float()).contiguous()
    return out


def _wn_pair(g, shape, fan_in, gain):
    """weight_v ~ N(0, gain^2/fan_in); weight_g = ||v|| (dim=0 norm) * (1 + 0.1 n)."""
    v = _n(g, *shape, std=gain / math.sqrt(fan_in))
    nrm = v.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], *([1] * (len(shape) - 1)))
    wg = nrm * _n(g, *nrm.shape, std=0.1, mean=1.0)
    return wg, v


def synth_generator_state(arch: VocoderArch, seed: int = DEFAULT_SEED + 1, folded: bool = False
                          ) -> "OrderedDict[str, torch.Tensor]":
    """Generator state dict (un-folded weight-norm form unless folded=True)."""
    g = _gen(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def put(name, shape, fan_in, gain, bias_n):
        wg, v = _wn_pair(g, shape, fan_in, gain)
        b = _n(g, bias_n, std=0.05)
        if folded:
            nrm = v.reshape(shape[0], -1).norm(dim=1).reshape(wg.shape)
            sd[f"{name}.bias"] = b
            sd[f"{name}.weight"] = v * (wg / nrm)
        else:
            sd[f"{name}.bias"] = b
            sd[f"{name}.weight_g"] = wg
            sd[f"{name}.weight_v"] = v

    C0 = arch.upsample_initial_channel
    put("conv_pre", (C0, arch.num_mels, 7), arch.num_mels * 7 * 9.0, 1.0, C0)   # mel values are O(5): damp
    for i, (u, k) in enumerate(zip(arch.upsample_rates, arch.upsample_kernel_sizes)):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        # ConvTranspose1d weight is (Cin, Cout, k); each output sees Cin*k/u taps
        put(f"ups.{i}", (cin, cout, k), cin * k / u, 1.0, cout)
    nk = len(arch.resblock_kernel_sizes)
    for i in range(len(arch.upsample_rates)):
        ch = C0 // (2 ** (i + 1))
        for j, (k, dil) in enumerate(zip(arch.resblock_kernel_sizes, arch.resblock_dilation_sizes)):
            if str(arch.resblock) == "2":                    # ResBlock2: `convs.<n>` only (I_ea/hifi_gan/models.py:56-61)
                for n in range(len(dil)):
                    put(f"resblocks.{i * nk + j}.convs.{n}", (ch, ch, k), ch * k, 1.0, ch)
                continue
            for n in range(len(dil)):
                put(f"resblocks.{i * nk + j}.convs1.{n}", (ch, ch, k), ch * k, 1.4, ch)
            for n in range(len(dil)):
                put(f"resblocks.{i * nk + j}.convs2.{n}", (ch, ch, k), ch * k, 0.8, ch)
    ch = C0 // (2 ** len(arch.upsample_rates))
    put("conv_post", (1, ch, 7), ch * 7, 0.2, 1)
    return sd


def synth_codebook(k: int = 100, dim: int = 80, seed: int = DEFAULT_SEED + 2) -> torch.Tensor:
    """(K, dim) centroids in log-mel range (mel_dump.py:31 clamps at log(1e-5) = -11.5)."""
    g = _gen(seed)
    return _n(g, k, dim, std=1.0, mean=-5.0)


def synth_wave(batch: int, n: int = 64000, seed: int = DEFAULT_SEED + 3, sr: int = 16000) -> torch.Tensor:
    """(B, n) speech-like clips: 5 harmonics of a gliding ~120 Hz fundamental + noise, amplitude 0.3."""
    out = torch.empty(batch, n, dtype=torch.float32)
    t = torch.arange(n, dtype=torch.float64) / sr
    for b in range(batch):
        g = _gen(seed + b)
        f0 = 100.0 + 60.0 * float(torch.rand(1, generator=g))
        glide = 20.0 + 40.0 * float(torch.rand(1, generator=g))
        phase = 2 * math.pi * (f0 * t + 0.5 * glide * t * t / max(t[-1].item(), 1e-9))
        x = torch.zeros(n, dtype=torch.float64)
        for hnum in range(1, 6):
            x += (1.0 / hnum) * torch.sin(hnum * phase + float(torch.rand(1, generator=g)) * 6.28318)
        env = 0.6 + 0.4 * torch.sin(2 * math.pi * 3.1 * t + float(torch.rand(1, generator=g)) * 6.28318)
        x = x * env
        x = 0.3 * x / x.abs().max().clamp_min(1e-9)
        x = x.float() + 0.01 * _n(g, n)
        out[b] = x
    return out


def synth_mel(batch: int, tm: int = 200, num_mels: int = 80, seed: int = DEFAULT_SEED + 4) -> torch.Tensor:
    """(B, num_mels, Tm) log-mel-like input: smooth in time, clamped at log(1e-5)."""
    g = _gen(seed)
    x = _n(g, batch, num_mels, tm + 4, std=1.5)
    k = torch.tensor([0.1, 0.2, 0.4, 0.2, 0.1]).view(1, 1, 5)
    x = torch.nn.functional.conv1d(x.reshape(batch * num_mels, 1, tm + 4), k).reshape(batch, num_mels, tm)
    tilt = torch.linspace(-3.0, -7.0, num_mels).view(1, num_mels, 1)
    return (x + tilt).clamp_min(math.log(1e-5)).contiguous()


def synth_mask_frames(batch: int, num_frames: int, mask_frames: int, seed: int = DEFAULT_SEED + 5,
                      frame_rate: float = 50.0) -> torch.Tensor:
    """(B,) int32 first masked frame, uniformly in [0.5 s, T - 0.7 s - mask] snapped to 20 ms frames
    (mirrors the random positions of I_ea/mask_pos_len.py:33-35)."""
    g = _gen(seed)
    lo = min(int(0.5 * frame_rate), max(num_frames - mask_frames - 1, 0))
    hi = max(num_frames - mask_frames - int(0.7 * frame_rate), lo + 1)
    return torch.randint(lo, hi, (batch,), generator=g, dtype=torch.int32)


def synth_f0_vqvae_state(desc, l_bins: int = 20, seed: int = 11) -> dict:
    """Random parameters with the names and shapes of the `FoVQVAE` checkpoint's encoder + bottleneck
    (I_da/src/model.py:16-21, jukebox.py:11-116, resnet.py:29-97, vq.py:15-22)."""
    g = torch.Generator().manual_seed(seed)
    k, _ = desc.down_kernel()
    sd = {}
    pre = "encoder.level_blocks.0.model."

    def conv(name, cout, cin, kk):
        sd[name + ".weight"] = torch.randn(cout, cin, kk, generator=g) * (1.0 / (cin * kk) ** 0.5)
        sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.1

    for i in range(desc.down_t):
        conv(f"{pre}{i}.0", desc.width, desc.in_width if i == 0 else desc.width, k)
        for j in range(desc.depth):
            conv(f"{pre}{i}.1.model.{j}.model.1", desc.n_state, desc.width, 3)
            conv(f"{pre}{i}.1.model.{j}.model.3", desc.width, desc.n_state, 1)
    conv(f"{pre}{desc.down_t}", desc.out_width, desc.width, 3)
    sd["vq.level_blocks.0.k"] = torch.randn(l_bins, desc.out_width, generator=g)
    return sd
