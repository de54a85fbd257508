"""Diagnostic: pairwise waveform RMS between fusion masks of the fp16 vocoder (SI_VOC_FUSE is read per context) and the
fp32 oracle, on truncated generators (1..4 upsample stages) to localise an arithmetic difference to one width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as R
from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch
from speech_inpainting_amd.engine import InpaintingEngine


def rms(a, b):
    return float((a.double() - b.double()).pow(2).mean().sqrt())


def eng(varch, gsd, mask, voc="fp16"):
    os.environ["SI_VOC_FUSE"] = str(mask)
    h = HubertArch.tiny()
    return InpaintingEngine(h, varch, 20, "cuda:0", "fp32", voc).load_state(synth.synth_hubert_state(h), gsd, synth.synth_codebook(20))


for rates, ks in (((8, 8), (16, 16)), ((8, 8, 2), (16, 16, 4)), ((8, 8, 2, 2), (16, 16, 4, 4))):
    varch = VocoderArch(upsample_rates=rates, upsample_kernel_sizes=ks)
    gsd = synth.synth_generator_state(varch)
    mel = synth.synth_mel(2, 57, 80, 77)
    ref = R.generator_forward(gsd, varch, mel)[:, 0, :]
    outs = {}
    for name, mask in (("none", 0), ("all", 1), ("narrow", 32 | 64), ("c128", 128), ("c256", 256)):
        outs[name] = eng(varch, gsd, mask).vocode(mel.cuda(), stretch=False).cpu()
    x3 = eng(varch, gsd, 0, "bf16x3").vocode(mel.cuda(), stretch=False).cpu()
    print(f"rates {rates}: signal rms {float(ref.pow(2).mean().sqrt()):.4f}; bf16x3 vs oracle {rms(x3, ref):.2e}")
    for k, v in outs.items():
        print(f"   {k:>7}: vs oracle {rms(v, ref):.3e}   vs none {rms(v, outs['none']):.3e}")
