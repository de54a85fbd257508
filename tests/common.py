"""Shared helpers for the parity tests: rebuild a golden case's synthetic inputs from its metadata."""
import json
import os

import numpy as np
import torch

from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    ha = dict(meta["harch"])
    for k in ("conv_dim", "conv_kernel", "conv_stride"):
        ha[k] = tuple(ha[k])
    va = dict(meta["varch"])
    for k in ("upsample_rates", "upsample_kernel_sizes", "resblock_kernel_sizes"):
        va[k] = tuple(va[k])
    va["resblock_dilation_sizes"] = tuple(tuple(d) for d in va["resblock_dilation_sizes"])
    harch, varch = HubertArch(**ha), VocoderArch(**va)
    seed = meta["seed"]
    case = dict(
        meta=meta, z=z, harch=harch, varch=varch,
        hsd=synth.synth_hubert_state(harch, seed, "legacy" if meta["legacy_pos"] else "parametrizations"),
        gsd=synth.synth_generator_state(varch, seed + 1),
        cb=synth.synth_codebook(meta["K"], 80, seed + 2),
        wave=synth.synth_wave(meta["B"], meta["N"], seed + 3),
        mel=synth.synth_mel(meta["B"], meta["Tm"], 80, seed + 4),
        frame_pos=[int(p) for p in z["frame_pos"]],
    )
    if "head_bias" in z.files:      # a fixture input: the head bias shifted so that the head output is centred on the masked frames
        case["hsd"]["final_layers.1.bias"] = torch.from_numpy(z["head_bias"]).clone()
    probe = np.asarray([float(case["hsd"]["final_layers.1.weight"][0, 0]), float(case["gsd"]["conv_post.weight_v"][0, 0, 0]),
                        float(case["cb"][0, 0]), float(case["wave"][0, 100]), float(case["mel"][0, 0, 0])])
    # the fixture is only meaningful if the seeded generator reproduces the tensors it was made from
    assert np.allclose(probe, z["weight_probe"], rtol=0, atol=1e-7), "synthetic RNG drifted from the golden fixtures"
    return case


def rms(a, b=None):
    a = torch.as_tensor(a).double()
    if b is not None:
        a = a - torch.as_tensor(b).double()
    return float(a.pow(2).mean().sqrt())
