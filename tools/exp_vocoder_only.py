"""Timing driver: the fp16 vocoder alone at the bench shape (B = 32, 200 mel frames -> 344), a few passes; run it under
rocprofv3 --kernel-trace and read tools/trace_respair.py.  SI_HIP_LIB selects a diagnostic build of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import InpaintingEngine

h, v = HubertArch.tiny(), VocoderArch.v1()
eng = InpaintingEngine(h, v, 20, "cuda:0", "fp32", "fp16").load_state(synth.synth_hubert_state(h), synth.synth_generator_state(v), synth.synth_codebook(20))
mel = synth.synth_mel(32, 200, 80, 5).cuda()
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(2):
    w = eng.vocode(mel, stretch=True)
torch.cuda.synchronize()
eng.ctx.profile_start(200 * passes)
for _ in range(passes):
    w = eng.vocode(mel, stretch=True)
torch.cuda.synchronize()
rows = eng.ctx.profile_stop()
print(f"vocoder alone, B=32: {sum(r['ms'] for r in rows) / passes:.3f} ms per pass in kernels ({os.environ.get('SI_HIP_LIB', 'default lib')}); checksum {float(w.double().abs().sum()):.9e}")
for r in sorted(rows, key=lambda r: -r["ms"]):
    print(f"  {r['name']:<28} {r['launches'] / passes:5.1f} x {1e3 * r['ms'] / r['launches']:8.1f} us = {r['ms'] / passes:7.3f} ms  {r['flops'] / r['ms'] / 1e9 if r['ms'] else 0:7.1f} TFLOP/s")
if "timeline" in os.environ.get("SI_HIP_LIB", ""):
    import ctypes
    out = (ctypes.c_ulonglong * 20)()
    eng.ctx.lib.si_debug_rpw_timeline(out, 1)
    names = ("tile staging", "conv-1 slabs", "phase-1 epilogue", "conv-2 slabs", "acc -> image", "output pass")
    for i, c in enumerate((128, 256)):
        v = out[10 * i:10 * i + 10]
        tot = sum(v[:6])
        if tot:
            print(f"C={c}: {v[6]} tiles on {v[7]} workgroups, {tot / v[6] * 0.01:.2f} us per tile: " + ", ".join(f"{n} {100.0 * x / tot:.1f} %" for n, x in zip(names, v[:6]))
                  + f"; of acc -> image: wave 0 waiting behind the last slab {100.0 * v[8] / tot:.1f} %, its image stores {100.0 * v[9] / tot:.1f} %")
    out = (ctypes.c_ulonglong * 10)()
    eng.ctx.lib.si_debug_rc_timeline(out, 1)
    names = ("tile staging", "c1", "c1 epilogue", "c2", "c2 epilogue", "output pass")
    tot = sum(out[:6])
    if tot:
        print(f"chain C=32: {out[8]} tiles on {out[9]} workgroups, {tot / out[8] * 0.01:.2f} us per tile: " + ", ".join(f"{n} {100.0 * x / tot:.1f} %" for n, x in zip(names, out[:6]))
              + f"; of the c2 epilogue, wave 0's own work / its wait at the closing barrier: {100.0 * out[6] / tot:.1f} / {100.0 * out[7] / tot:.1f} %")
