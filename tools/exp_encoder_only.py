"""Timing driver: the bf16 encoder alone at the bench shape (B = 32 x 4 s, HuBERT-base), per GEMM shape.  Run with
SI_PROF_SHAPES=1 so that every launch of the encoder's GEMM kernels is listed under its own (M, N, K); SI_HIP_LIB selects an
A/B build of the library.  usage: python tools/exp_encoder_only.py [passes] [base|large]"""
import os, sys
os.environ.setdefault("SI_PROF_SHAPES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import InpaintingEngine

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
h = HubertArch.large() if (len(sys.argv) > 2 and sys.argv[2] == "large") else HubertArch.base()
v = VocoderArch.tiny()
eng = InpaintingEngine(h, v, 100, "cuda:0", "bf16", "fp16").load_state(synth.synth_hubert_state(h), synth.synth_generator_state(v), synth.synth_codebook(100))
B = 32 if h.hidden_size == 768 else 16
wave = synth.synth_wave(B, 64000, 3).cuda()
for _ in range(3):
    f = eng.encode(wave)
torch.cuda.synchronize()
eng.ctx.profile_start(400 * passes)
for _ in range(passes):
    f = eng.encode(wave)
torch.cuda.synchronize()
rows = eng.ctx.profile_stop()
tot = sum(r["ms"] for r in rows) / passes
print(f"encoder alone, B={B}: {tot:.3f} ms per pass in kernels ({os.environ.get('SI_HIP_LIB', 'default lib')})")
gemm = 0.0
for r in sorted(rows, key=lambda r: -r["ms"]):
    n = r["launches"] / passes
    if "gemm" in r["name"]:
        gemm += r["ms"] / passes
    print(f"  {r['name']:<44} {n:5.1f} x {1e3 * r['ms'] / r['launches']:8.1f} us = {r['ms'] / passes:7.3f} ms  {r['flops'] / r['ms'] / 1e9 if r['ms'] else 0:7.1f} TFLOP/s")
print(f"  GEMM kernels together: {gemm:.3f} ms per pass; checksum {float(f.double().abs().sum()):.6e}")
if "timeline" in os.environ.get("SI_HIP_LIB", ""):
    import ctypes
    out = (ctypes.c_ulonglong * 24)()
    eng.ctx.lib.si_debug_g256_timeline(out, 1)
    for c, nm in enumerate(("conv K=1536", "N=2304 (QKV)", "N=3072 (FFN1)", "other")):
        v = out[6 * c:6 * c + 6]
        if v[4]:
            print(f"  gemm256 timeline, {nm}: {v[3]} tiles on {v[4]} workgroup launches; per workgroup: prologue {v[0] / v[4] * 0.01:.2f} us, "
                  f"K loops {v[1] / v[4] * 0.01:.2f} us ({v[1] / max(v[3], 1) * 0.01:.2f} per tile), epilogues {v[2] / v[4] * 0.01:.2f} us "
                  f"({v[2] / max(v[3], 1) * 0.01:.2f} per tile), whole kernel {v[5] / v[4] * 0.01:.2f} us")
    out = (ctypes.c_ulonglong * 16)()
    if hasattr(eng.ctx.lib, "si_debug_gcu_timeline") and eng.ctx.lib.si_debug_gcu_timeline(out, 1) == 0:
        for c, nm in enumerate(("N=3072 (FFN1)", "N=2304 (QKV)", "N=768 K=3072 (FFN2)", "other")):
            v = out[4 * c:4 * c + 4]
            if v[3]:
                print(f"  gemmcu timeline, {nm}: {v[3]} workgroups; per workgroup: prologue {v[0] / v[3] * 0.01:.2f} us, K loop {v[1] / v[3] * 0.01:.2f} us, "
                      f"epilogue (stores drained) {v[2] / v[3] * 0.01:.2f} us")
