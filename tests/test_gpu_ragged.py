"""BASELINE configs[4]: clips of DIFFERENT lengths sharing every launch (the library's ragged-batch entry points).

The reference runs one file of any length per invocation (I_ea/predict.py:76-207), so the contract of a ragged batch is:
every clip's result equals that clip run ALONE -- bit for bit in every arithmetic mode (same kernels' arithmetic, the
clip's own statistics / lengths / padding), and therefore equal to the oracle's single-clip result in fp32.
"""
import numpy as np
import pytest
import torch

from tests.common import rms

pytestmark = pytest.mark.gpu


def _mk(harch, varch, enc="fp32", voc="fp32", K=100):
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.engine import InpaintingEngine
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(K)
    return InpaintingEngine(harch, varch, K, "cuda:0", enc, voc).load_state(hsd, gsd, cb), (hsd, gsd, cb)


def _clips(secs, seed0=100):
    """Per clip: 16 kHz and 22.05 kHz synthetic waves of `secs[i]` seconds (the 22.05 kHz length as the resampler rounds it)."""
    from speech_inpainting_amd import synth
    w16 = [synth.synth_wave(1, int(round(s * 16000)), seed0 + i)[0].numpy() for i, s in enumerate(secs)]
    w22 = [synth.synth_wave(1, -(-len(w) * 441 // 320), seed0 + 500 + i, sr=22050)[0].numpy() for i, w in enumerate(w16)]
    return w16, w22


def _u410(n, seed=1234):
    """lengths drawn U[4 s, 10 s], seed 1234."""
    g = torch.Generator().manual_seed(seed)
    return (4.0 + 6.0 * torch.rand(n, generator=g)).tolist()


def _assert_same(a, b, what):
    for k in ("feats", "labels", "mel", "wave"):
        x, y = a[k], b[k]
        assert x.shape == y.shape, (what, k, tuple(x.shape), tuple(y.shape))
        assert torch.equal(x, y), (what, k, float((x.double() - y.double()).abs().max()))


@pytest.mark.parametrize("enc,voc", [("fp32", "fp32"), ("bf16", "fp16"), ("bf16", "bf16x3"), ("bf16", "bf16")])
def test_ragged_batch_equals_single_clip_runs_tiny(enc, voc):
    """Tiny architectures, every arithmetic mode: one ragged call over clips of 0.9 ... 2.6 s (sub-batches of 4 and one batch of
    all 7; masked and blind) against each clip run alone through the uniform entry points: bit-identical."""
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.predict import predict_ragged
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    eng, _ = _mk(harch, varch, enc, voc, K=50)
    secs = [1.0, 2.6, 1.5, 0.9, 2.2, 1.5, 1.85]
    w16, w22 = _clips(secs)
    pos, lm = [10, 30, 20, 5, 55, 40, 33], 5
    for blind in (False, True):
        alone = predict_ragged(eng, w16, w22, pos, lm, blind=blind, exact_length=True, max_batch=1)
        for mb in (4, 32):
            got = predict_ragged(eng, w16, w22, pos, lm, blind=blind, max_batch=mb)
            for i in range(len(secs)):
                _assert_same(got[i], alone[i], (enc, voc, blind, mb, i))


def test_ragged_32_clips_bench_arithmetic_equals_single_clip_runs():
    """configs[4] at the bench's arithmetic (HuBERT-base bf16 + HiFi-GAN V1 fp16 stream): 32 clips, lengths U[4 s, 10 s] seed
    1234, ONE predict_ragged call (one batch: every launch shared) -> every clip bit-identical to that clip alone.  Clips
    shorter than 5.14 s have T <= 256 frames and run the whole-K/V attention kernel alone but the tiled one in the batch: the
    two are bit-identical by construction, and this test is where that is checked.  Blind mode on the 8 shortest + longest."""
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.predict import plan_ragged_batches, predict_ragged, storage_padding
    harch, varch = HubertArch.base(), VocoderArch.v1()
    eng, _ = _mk(harch, varch, "bf16", "fp16")
    secs = _u410(32)
    assert min(secs) < 5.0 and max(secs) > 9.0
    w16, w22 = _clips(secs)
    frames = [harch.num_frames(len(w)) for w in w16]
    assert min(frames) <= 256 < max(frames)
    g = torch.Generator().manual_seed(77)
    lm = 10
    pos = [int(torch.randint(5, f - lm - 5, (1,), generator=g)) for f in frames]
    plan = plan_ragged_batches([len(w) for w in w16], 32)
    assert len(plan) == 1 and len(plan[0]) == 32
    print(f"32 ragged clips: {sum(secs):.1f} s of audio, storage padding {storage_padding([len(w) for w in w16], plan):.3f} in one batch, "
          f"{storage_padding([len(w) for w in w16], plan_ragged_batches([len(w) for w in w16], 32, 0.15)):.3f} in sub-batches of <= 15 %")
    got = predict_ragged(eng, w16, w22, pos, lm, max_batch=32)
    alone = predict_ragged(eng, w16, w22, pos, lm, exact_length=True, max_batch=1)
    for i in range(32):
        _assert_same(got[i], alone[i], ("masked", i, secs[i]))
    order = sorted(range(32), key=lambda i: secs[i])
    sub = order[:4] + order[-4:]
    gb = predict_ragged(eng, [w16[i] for i in sub], [w22[i] for i in sub], [0] * 8, 0, blind=True, max_batch=8)
    ab = predict_ragged(eng, [w16[i] for i in sub], [w22[i] for i in sub], [0] * 8, 0, blind=True, exact_length=True, max_batch=1)
    for k in range(8):
        assert gb[k]["labels"].shape[1] == min(frames[sub[k]], gb[k]["mel"].shape[2])
        _assert_same(gb[k], ab[k], ("blind", sub[k]))


def test_ragged_fp32_matches_oracle_on_sampled_clips():
    """The same 32 lengths in fp32: shortest, median and longest clip of the ragged batch against the ORACLE run on that clip
    alone (mel front-end included): labels exact, waveform <= 1e-4 RMS -- masked, and blind on the shortest / longest."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.predict import predict_ragged
    harch, varch = HubertArch.base(), VocoderArch.v1()
    eng, (hsd, gsd, cb) = _mk(harch, varch)
    secs = _u410(32)
    w16, w22 = _clips(secs)
    frames = [harch.num_frames(len(w)) for w in w16]
    g = torch.Generator().manual_seed(77)
    lm = 10
    pos = [int(torch.randint(5, f - lm - 5, (1,), generator=g)) for f in frames]
    order = sorted(range(32), key=lambda i: secs[i])
    torch.set_num_threads(16)
    got = predict_ragged(eng, w16, w22, pos, lm, max_batch=32)
    for i in (order[0], order[16], order[-1]):
        s22, e22 = pos[i] * 320 * 22050 // 16000, (pos[i] + lm) * 320 * 22050 // 16000
        mel = R.masked_mel([w22[i]], [s22], [e22])
        ref = R.predict_batch(hsd, harch, gsd, varch, cb, torch.from_numpy(w16[i])[None], mel, [pos[i]], lm)
        assert got[i]["wave"].shape == tuple(ref["wave"].shape)
        assert torch.equal(got[i]["labels"].cpu(), ref["labels"]), ("masked", i)
        e = rms(got[i]["wave"].cpu(), ref["wave"])
        print(f"ragged fp32 clip {i} ({secs[i]:.2f} s, masked): waveform rms error {e:.3e} vs the oracle")
        assert e <= 1e-4 and rms(got[i]["feats"].cpu(), ref["feats"]) <= 1e-4 * rms(ref["feats"])
    sub = [order[0], order[-1], order[7]]
    gb = predict_ragged(eng, [w16[i] for i in sub], [w22[i] for i in sub], [0, 0, 0], 0, blind=True, max_batch=8)
    for k in (0, 1):
        i = sub[k]
        ref = R.predict_batch(hsd, harch, gsd, varch, cb, torch.from_numpy(w16[i])[None], R.masked_mel([w22[i]], None, None), [0], 0, blind=True)
        assert gb[k]["wave"].shape == tuple(ref["wave"].shape)
        agree = float((gb[k]["labels"].cpu() == ref["labels"]).float().mean())
        e = rms(gb[k]["wave"].cpu(), ref["wave"])
        print(f"ragged fp32 clip {i} ({secs[i]:.2f} s, blind): label agreement {agree:.4f} over {ref['labels'].numel()} frames, rms error {e:.3e}")
        assert agree == 1.0 and e <= 1e-4


def test_bf16_tiled_attention_kernel_at_T499():
    """`attention_bf16in_kernel` (T > 256: every clip longer than 5.14 s) in the bf16 encoder on a 10 s clip: head output
    within 2e-2 relative of the fp32 ORACLE, labels of a second run identical, and bit-identical across two runs."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    harch, varch = HubertArch.base(), VocoderArch.v1()
    eng, (hsd, gsd, cb) = _mk(harch, varch, "bf16", "fp16")
    N = 160000
    assert harch.num_frames(N) == 499
    wave = synth.synth_wave(2, N, 31)
    ms = torch.tensor([200 * 320 + 80, 350 * 320 + 80], dtype=torch.int32)
    ml = torch.full((2,), 10 * 320 - 81, dtype=torch.int32)
    a = eng.encode(wave.cuda(), ms.cuda(), ml.cuda())
    b = eng.encode(wave.cuda(), ms.cuda(), ml.cuda())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.custom_model_forward(hsd, harch, R.mask_and_normalize(wave, ms.tolist(), ml.tolist()))
    rel = rms(a.cpu(), ref) / rms(ref)
    print(f"bf16 encoder, T = 499 (tiled attention kernel): head output relative rms error {rel:.3e} vs the fp32 oracle")
    assert rel <= 2e-2


@pytest.mark.parametrize("H,heads", [(768, 12), (1024, 16)])
def test_posconv_kernel_against_a_float64_convolution_of_the_same_bf16_operands(H, heads):
    """The dedicated positional-conv kernel of the bf16 encoder (posconv.hip: 48 / 64 channels per group) and the generic tap-GEMM
    (SI_ENC_POSCONV=0, read when a context is created), each against a float64 grouped convolution of the SAME bf16-rounded operands
    (the captured projection output, the folded weights): h + GELU(conv(h) + b) within fp32 accumulation error for both -- on 4 s
    clips (two clips per workgroup), 10 s clips (512-row blocks), 10.6 s clips (two blocks per clip); then a ragged batch (packed
    rows), where every clip must still be bit-identical to that clip alone."""
    import os
    import torch.nn.functional as F
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    # pre-LN ("stable") flavour: the tap `encoder_in` is then the positional conv's output itself, h + gelu(conv(h) + b)
    harch = HubertArch.tiny(hidden_size=H, num_attention_heads=heads, num_hidden_layers=1, intermediate_size=512,
                            num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, do_stable_layer_norm=True)
    varch = VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)

    def make():
        return InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp16").load_state(hsd, gsd, cb)
    new = make()
    os.environ["SI_ENC_POSCONV"] = "0"
    try:
        old = make()
    finally:
        os.environ.pop("SI_ENC_POSCONV", None)
    w = R._conv_weight(hsd, "base_model.encoder.pos_conv_embed.conv", dim=2).to(torch.bfloat16).double()
    bias = hsd["base_model.encoder.pos_conv_embed.conv.bias"].double()
    torch.set_num_threads(16)
    for B, N in ((3, 64000), (2, 160000), (1, 170000)):
        wave = synth.synth_wave(B, N, 13).cuda()
        T = harch.num_frames(N)
        res = {}
        for name, eng in (("posconv", new), ("tapgemm", old)):
            caps = eng.ctx.capture(["projected", "encoder_in"], capacity=B * T * H)
            eng.encode(wave)
            torch.cuda.synchronize()
            res[name] = {k: v.cpu().reshape(B, T, H) for k, v in caps.items()}
            eng.ctx.clear_captures()
        assert torch.equal(res["posconv"]["projected"], res["tapgemm"]["projected"])
        h = res["posconv"]["projected"]
        conv = F.conv1d(h.to(torch.bfloat16).double().transpose(1, 2), w, bias, padding=64, groups=16)[:, :, :-1].transpose(1, 2)
        ref = h.double() + F.gelu(conv)
        e_new, e_old = rms(res["posconv"]["encoder_in"], ref) / rms(ref), rms(res["tapgemm"]["encoder_in"], ref) / rms(ref)
        print(f"H={H} B={B} T={T}: h + gelu(posconv(h)) vs float64 on the same bf16 operands: posconv.hip {e_new:.2e}, tap-GEMM {e_old:.2e} relative")
        d = rms(res["posconv"]["encoder_in"], res["tapgemm"]["encoder_in"]) / rms(ref)
        print(f"        the two kernels differ by {d:.2e} relative (fp32 sum order)")
        assert e_new <= 2e-5 and e_old <= 2e-5 and abs(e_new - e_old) <= 1e-6 and d <= 1e-6
    lens = [64000, 100000, 52345, 160000, 80000]
    wave = synth.synth_wave(len(lens), max(lens), 14)
    got = new.encode_ragged(wave.cuda(), lens)
    for i, n in enumerate(lens):
        alone = new.encode(wave[i:i + 1, :n].contiguous().cuda())
        assert torch.equal(got[i, :alone.shape[1]], alone[0]), (H, i)
