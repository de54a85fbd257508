"""The non-default code paths behind the library's environment knobs stay parity-green.

The knobs are read once per process, so each combination runs in its own (sequential) subprocess: the tiny golden
case through the C ABI with a bf16 encoder + fp16 vocoder and with the all-fp32 path, against the stored reference
waveform.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.common import load_case, rms
from tests.test_gpu_parity import _engine, _run
out = {}
c = load_case("tiny_group")
z = c["z"]
for tag, enc, voc in (("fp32", "fp32", "fp32"), ("fp16", "fp32", "fp16"), ("bf16", "fp32", "bf16"), ("x3", "fp32", "bf16x3")):
    o = _run(_engine(c, enc=enc, voc=voc), c)
    out[tag] = {"err": float(rms(o["wave"], z["wave"])), "labels": bool(np.array_equal(o["labels"].numpy(), z["labels"]))}
o = _run(_engine(c, enc="bf16", voc="fp16"), c)
fr = torch.from_numpy(z["feats"])
out["enc_bf16"] = {"feat_rel": float(rms(o["feats"], fr) / rms(fr))}
print("RESULT " + json.dumps(out))
""" % ROOT

COMBOS = [
    {"SI_VOC_RES16": "0"},                                   # fp16 operands, fp32 residual stream
    {"SI_VOC_FUSE": "0"},                                    # fp16 stream with the conv pairs as two launches
    {"SI_VOC_OPREADY": "0", "SI_ENC_OPREADY": "0"},         # consumers convert fp32 activations while staging
    {"SI_ATT_BF16": "0"},                                    # bf16 encoder with the exact-fp32 attention kernel
    {"SI_VOC_RES16": "0", "SI_VOC_OPREADY": "0", "SI_ENC_OPREADY": "0", "SI_ATT_BF16": "0"},   # every non-default arithmetic path at once
    {"SI_ENC_LINGEMM": "0"},                                 # encoder GEMMs on the generic tap-GEMM
    {"SI_VOC_CHAIN": "0"},                                   # C = 32 stage as one launch per conv pair instead of per resblock
]


@pytest.mark.parametrize("combo", COMBOS, ids=lambda c: ",".join(f"{k}={v}" for k, v in c.items()))
def test_knob_combination_stays_parity_green(combo):
    env = dict(os.environ)
    env.update(combo)
    p = subprocess.run([sys.executable, "-c", SNIPPET], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    print(combo, r)
    assert r["fp32"]["labels"] and r["fp32"]["err"] <= 1e-5
    assert r["x3"]["labels"] and r["x3"]["err"] <= 1e-5
    assert r["fp16"]["labels"] and r["fp16"]["err"] <= 2e-4
    assert r["bf16"]["labels"] and r["bf16"]["err"] <= 1e-3
    assert r["enc_bf16"]["feat_rel"] <= 2e-2
