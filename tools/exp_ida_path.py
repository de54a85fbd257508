"""Timing driver: I_da's inpainting() path at the configs[3] per-GPU shape (HuBERT-large, 16 clips x 4 s, 400 ms mask, K units,
unit HiFi-GAN): per-kernel-family time of one engine.ida_inpaint_batch call.  usage: python tools/exp_ida_path.py [passes] [enc dtype] [voc dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from speech_inpainting_amd import native, synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import CodeGenerator, F0Quantizer, InpaintingEngine

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
enc = sys.argv[2] if len(sys.argv) > 2 else "bf16"
voc = sys.argv[3] if len(sys.argv) > 3 else "fp16"
harch = HubertArch.large()
varch = VocoderArch(upsample_rates=(5, 4, 4, 2, 2), upsample_kernel_sizes=(11, 8, 8, 4, 4), upsample_initial_channel=512, num_mels=384, sampling_rate=16000)
B, N, K, L = 16, 64000, 100, 18
hsd = {k: v for k, v in synth.synth_hubert_state(harch, 77).items() if k.startswith("base_model.")}
eng = InpaintingEngine(harch, varch, K, "cuda:0", enc, voc).load_state(hsd, synth.synth_generator_state(varch, 78))
g = torch.Generator().manual_seed(1)
gen = CodeGenerator(eng, torch.randn(K, 128, generator=g) * 0.5, torch.randn(20, 128, generator=g) * 0.5,
                    f0_quantizer=F0Quantizer(eng, synth.synth_f0_vqvae_state(native.F0EncDesc(), 20, seed=79)))
wave = synth.synth_wave(B, N, 81).cuda()
cent = torch.randn(K, harch.hidden_size, generator=g).cuda()
f0 = torch.randn(B, 1, N // 80 - 3, generator=g).cuda()
spk = (torch.randn(B, 128, generator=g) * 0.5).cuda()
for _ in range(2):
    out = eng.ida_inpaint_batch(wave, 24000, 6400, cent, gen, f0, spk, output_layer=L)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    out = eng.ida_inpaint_batch(wave, 24000, 6400, cent, gen, f0, spk, output_layer=L)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / passes
eng.ctx.profile_start(1500 * passes)
for _ in range(passes):
    out = eng.ida_inpaint_batch(wave, 24000, 6400, cent, gen, f0, spk, output_layer=L)
torch.cuda.synchronize()
rows = eng.ctx.profile_stop()
tot = sum(r["ms"] for r in rows) / passes
print(f"I_da inpainting(), B={B} x 4 s, HuBERT-large layer {L}, K={K}, encoder {enc}, vocoder {voc}: {1e3 * wall:.2f} ms wall per call "
      f"({B * 4.0 / wall:.0f} x real-time), {tot:.2f} ms in kernels; two waveforms {tuple(out['audio_inp'].shape)}")
for r in sorted(rows, key=lambda r: -r["ms"])[:16]:
    print(f"  {r['name']:<28} {r['launches'] / passes:6.1f} x {1e3 * r['ms'] / r['launches']:8.1f} us = {r['ms'] / passes:7.3f} ms")
