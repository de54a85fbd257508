"""Diagnostic: per-phase cycle shares of the tap-GEMM conv loop (needs a -DTG_STAMPS build of the library, see
the TG_STAMPS block in csrc/tapgemm.hip; usage: SI_HIP_LIB=<stamps .so> python tools/exp_stamps.py [voc dtype])."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_inpainting_amd import synth, native
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import InpaintingEngine
voc = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
eng = InpaintingEngine(HubertArch.tiny(), VocoderArch.v1(), 100, "cuda:0", "fp32", voc)
eng.load_state(synth.synth_hubert_state(HubertArch.tiny()), synth.synth_generator_state(VocoderArch.v1()), synth.synth_codebook())
lib = native.load_library()
mel = synth.synth_mel(32, 200).cuda()
eng.vocode(mel); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 24)()
lib.si_debug_stamps(buf, 1)
eng.vocode(mel); torch.cuda.synchronize()
lib.si_debug_stamps(buf, 0)
names = ["issue", "compute", "land(wait+LDS write)", "barrier", "prologue", "epilogue", "total", "waves"]
for fam, label in enumerate(("N-tile 128 (C>=128 stages, ups, conv_pre)", "N-tile 64 (C=64 stage)", "N-tile 32 (C=32 stage)")):
    v = buf[fam * 8:(fam + 1) * 8]
    if not v[7]:
        continue
    print(f"-- {label}: {v[7]} waves, {v[6] / v[7]:.0f} cycles per wave")
    for n, x in zip(names[:6], v[:6]):
        print(f"   {n:22s} {100.0 * x / v[6]:6.1f} %   {x / v[7]:10.0f} cycles/wave")

if hasattr(lib, "si_debug_stamps_pp"):
    b2 = (ctypes.c_ulonglong * 16)()
    lib.si_debug_stamps_pp(b2, 1)
    eng.vocode(mel); torch.cuda.synchronize()
    lib.si_debug_stamps_pp(b2, 0)
    pn = ["compute phases", "stage phases", "barrier after compute", "barrier after stage", "prologue", "epilogue"]
    for grp in (0, 1):
        v = b2[grp * 8:(grp + 1) * 8]
        if not v[7]:
            continue
        print(f"-- ping-pong kernel, wave group {grp}: {v[7]} waves, {v[6] / v[7]:.0f} cycles per wave")
        for n, x in zip(pn, v[:6]):
            print(f"   {n:22s} {100.0 * x / v[6]:6.1f} %   {x / v[7]:10.0f} cycles/wave")
