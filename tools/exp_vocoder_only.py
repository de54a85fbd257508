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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    w = eng.vocode(mel, stretch=True)
torch.cuda.synchronize()
print("ok", float(w.float().abs().mean()))
if "ablate" in os.environ.get("SI_HIP_LIB", ""):
    import ctypes
    lib = eng.ctx.lib
    out = (ctypes.c_ulonglong * 8)()
    lib.si_debug_rpw_stamps(out, 1)
    for i, c in enumerate((128, 256)):
        dt, dr, n, slabs = out[4 * i:4 * i + 4]
        if n:
            print(f"C={c}: in-loop clock {dt / dr * 0.1:.3f} GHz; {dt / slabs:.0f} shader cycles per slab ({slabs / n:.1f} slabs per workgroup on average), {n} workgroups")
