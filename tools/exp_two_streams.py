"""Experiment: the bf16 encoder at the bench shape as ONE batch of 32 on one stream against TWO half batches of 16 on two streams
(two engines: separate workspaces), wall clock.  One-round launches leave every kernel boundary exposed (the slowest workgroup, the
end-of-kernel write-back, the next launch's ramp); two independent streams can fill each other's boundaries -- if the hardware runs
their workgroups side by side.  usage: python tools/exp_two_streams.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import InpaintingEngine

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
h, v = HubertArch.base(), VocoderArch.tiny()
hsd, gsd, cb = synth.synth_hubert_state(h), synth.synth_generator_state(v), synth.synth_codebook(100)
engs = [InpaintingEngine(h, v, 100, "cuda:0", "bf16", "fp16").load_state(hsd, gsd, cb) for _ in range(2)]
wave = synth.synth_wave(32, 64000, 3).cuda()
halves = [wave[:16].contiguous(), wave[16:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def one():
    return engs[0].encode(wave)


def two():
    outs = []
    cur = torch.cuda.current_stream()
    for e, w, s in zip(engs, halves, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(e.encode(w))
    for s in streams:
        cur.wait_stream(s)
    return outs


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes * 1e3


for rep in range(2):
    a = timed(one)
    b = timed(two)
    print(f"encoder, B = 32: one stream {a:.3f} ms per pass; two half batches on two streams {b:.3f} ms")
f1 = one()
f2 = torch.cat(two(), 0)
torch.cuda.synchronize()
print("equal:", torch.equal(f1, f2))
