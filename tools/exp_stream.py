"""Timing driver for the request front (stream.py): host phases and per-kernel table of one configs[1] request."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import InpaintingEngine
from speech_inpainting_amd.stream import Request, RequestFront
harch, varch = HubertArch.base(), VocoderArch.v1()
B, n22 = 32, 88200
eng = InpaintingEngine(harch, varch, 100, "cuda:0", "bf16", "fp16").load_state(synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(100))
clips = list(synth.synth_wave(B, n22, 7, sr=22050).numpy())
pos = synth.synth_mask_frames(B, 199, 10, 8).tolist()
rq = Request(clips, pos, 10)
front = RequestFront(eng, 22050, 2)
for _ in front.run([rq] * 3): pass
torch.cuda.synchronize()
# host phases
s = front.slots[0]
t0 = time.perf_counter(); front._submit(s, rq); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter(); r = front._collect(s); t3 = time.perf_counter()
print(f"submit (host) {1e3*(t1-t0):.2f} ms, GPU drain {1e3*(t2-t1):.2f} ms, collect {1e3*(t3-t2):.2f} ms")
for steps in (10,):
    t0 = time.perf_counter()
    for res in front.run([rq] * steps): pass
    print(f"pipelined: {1e3*(time.perf_counter()-t0)/steps:.2f} ms per request")
eng.ctx.profile_filter(None); eng.ctx.profile_start(4000)
for _ in front.run([rq] * 2): pass
torch.cuda.synchronize()
rows = eng.ctx.profile_stop()
for r in sorted(rows, key=lambda r: -r["ms"])[:8]:
    print(f"  {r['name']:<28} {r['launches']/2:6.1f} x {1e3*r['ms']/r['launches']:8.1f} us = {r['ms']/2:7.3f} ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
front._submit(s, rq)
pr.disable()
torch.cuda.synchronize(); front._collect(s)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
