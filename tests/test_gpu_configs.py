"""The remaining BASELINE.json configs at their own shapes, against the (pinned) CPU oracle and through
size-independent properties."""
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


def test_config5_blind_10s_clip_matches_oracle():
    """configs[4]: blind inpainting (every frame replaced) on a 10 s clip: T = 499 encoder frames (attention streams
    16 key tiles), 500 mel frames -> 861 -> 220 416 samples."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames
    harch, varch = HubertArch.base(), VocoderArch.v1()
    eng, (hsd, gsd, cb) = _mk(harch, varch)
    N = 160000
    Tm = mel_frames(N * 22050 // 16000)
    assert harch.num_frames(N) == 499 and Tm == 500
    wave, mel = synth.synth_wave(1, N, 21), synth.synth_mel(1, Tm, 80, 22)
    pos = torch.zeros(1, dtype=torch.int32)
    out = eng.predict_batch(wave.cuda(), mel.cuda(), pos.cuda(), 0, blind=True)
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    ref = R.predict_batch(hsd, harch, gsd, varch, cb, wave, mel, [0], 0, blind=True)
    assert out["wave"].shape == (1, 861 * 256) == tuple(ref["wave"].shape)
    agree = float((out["labels"].cpu() == ref["labels"]).float().mean())
    err = rms(out["wave"].cpu(), ref["wave"])
    print(f"10 s blind: label agreement {agree:.3f} over {ref['labels'].numel()} frames, waveform rms error {err:.3e}")
    assert rms(out["feats"].cpu(), ref["feats"]) <= 1e-4 * rms(ref["feats"])
    assert agree == 1.0 and err <= 1e-4


def test_config4_large_encoder_batch_matches_oracle_per_clip():
    """configs[3] encoder: HuBERT-large (24 pre-LN layers, LayerNorm feature extractor with bias), 400 ms mask; a batch of
    3 clips with different mask positions against the oracle run clip by clip."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    harch, varch = HubertArch.large(), VocoderArch.tiny()
    eng, (hsd, gsd, cb) = _mk(harch, varch)
    B, N, lm = 3, 64000, 20
    wave = synth.synth_wave(B, N, 31)
    pos = torch.tensor([10, 90, 170], dtype=torch.int32)
    sl = [R.mask_samples_from_frames(int(p), lm) for p in pos]
    ms = torch.tensor([s for s, _ in sl], dtype=torch.int32)
    ml = torch.tensor([l for _, l in sl], dtype=torch.int32)
    feats = eng.encode(wave.cuda(), ms.cuda(), ml.cuda()).cpu()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.custom_model_forward(hsd, harch, R.mask_and_normalize(wave, ms.tolist(), ml.tolist()))
    assert feats.shape == ref.shape == (B, 199, 80)
    assert rms(feats, ref) <= 1e-4 * rms(ref)


def test_full_size_batch_is_clipwise_identical_to_single_clip_runs():
    """At the bench's own shape (B = 32, base + V1) in the bench's own arithmetic (bf16 encoder, fp16 vocoder with the fp16
    activation stream and the fused ResBlock kernels): every clip of the batch equals that clip run alone, bit for bit,
    two runs are identical -- sharding by utterance cannot change results -- and the fp16 waveform of clips 0 / 17 / 31
    stays within 2e-4 RMS of the fp32-equivalent bf16x3 vocoder on the same clip (chunking x fused kernels x 16-bit
    accumulate at full size)."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames
    harch, varch = HubertArch.base(), VocoderArch.v1()
    eng, _ = _mk(harch, varch, enc="bf16", voc="fp16")
    B, N, lm = 32, 64000, 10
    Tm = mel_frames(N * 22050 // 16000)
    wave, mel = synth.synth_wave(B, N, 41).cuda(), synth.synth_mel(B, Tm, 80, 42).cuda()
    pos = synth.synth_mask_frames(B, 199, lm, 43).cuda()
    a = eng.predict_batch(wave, mel, pos, lm)
    b = eng.predict_batch(wave, mel, pos, lm)
    assert torch.equal(a["wave"], b["wave"]) and torch.equal(a["labels"], b["labels"])
    assert a["wave"].shape == (B, 344 * 256) and bool(torch.isfinite(a["wave"]).all())
    for i in (0, 17, 31):
        one = eng.predict_batch(wave[i:i + 1].contiguous(), mel[i:i + 1].contiguous(), pos[i:i + 1].contiguous(), lm)
        assert torch.equal(one["labels"], a["labels"][i:i + 1])
        assert torch.equal(one["wave"], a["wave"][i:i + 1])
    ref_eng, _ = _mk(harch, varch, enc="bf16", voc="bf16x3")
    for i in (0, 17, 31):
        r = ref_eng.predict_batch(wave[i:i + 1].contiguous(), mel[i:i + 1].contiguous(), pos[i:i + 1].contiguous(), lm)
        assert torch.equal(r["labels"], a["labels"][i:i + 1])                      # same encoder arithmetic
        err = rms(a["wave"][i].cpu(), r["wave"][0].cpu())
        print(f"clip {i}: fp16 vs bf16x3 vocoder waveform rms {err:.3e}")
        assert err <= 2e-4
    # splice property: outside the masked frames the mel is untouched, inside it is a codebook row
    cb = synth.synth_codebook(100).cuda()
    for i in (3, 20):
        p = int(pos[i])
        assert torch.equal(a["mel"][i, :, :p], mel[i, :, :p]) and torch.equal(a["mel"][i, :, p + lm:], mel[i, :, p + lm:])
        assert torch.allclose(a["mel"][i, :, p:p + lm].T, cb[a["labels"][i]], atol=1e-6)


def test_vocoder_is_shift_consistent():
    """Domain property of a fully convolutional generator: away from the edges (receptive field ~ 3 mel frames of
    conv_pre + the MRF halos), delaying the mel by one frame delays the waveform by exactly `hop` samples."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    eng, _ = _mk(HubertArch.tiny(), VocoderArch.v1(), voc="bf16x3")
    mel = synth.synth_mel(1, 120, 80, 51).cuda()
    shifted = torch.roll(mel, 1, dims=2)
    a = eng.vocode(mel, stretch=False)[0]
    b = eng.vocode(shifted, stretch=False)[0]
    lo, hi = 40 * 256, 80 * 256
    assert rms(b[lo + 256:hi + 256].cpu(), a[lo:hi].cpu()) <= 1e-5


@pytest.mark.parametrize("voc", ["fp32", "bf16x3", "fp16", "bf16"])
def test_windowed_generator_passes_are_bit_identical_to_full_passes(voc):
    """The script runs the generator three times (masked, expected, inpainted mel: I_ea/predict.py:123-128,196-207); the mels differ
    only in the Lm spliced frames, so `predict_resident(diagnostics=True)` runs ONE full pass and the other two over the window the
    spliced frames can reach.  Against three full passes: bit-identical, in every vocoder mode, with masks at the start, in the
    middle and at the end of a clip (windows clamped at a clip edge keep that edge's real zero padding)."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.predict import predict_resident
    harch, varch = HubertArch.tiny(), VocoderArch.v1()
    eng, _ = _mk(harch, varch, "fp32", voc)
    B, n16 = 3, 48000
    n22 = n16 * 441 // 320
    wave = synth.synth_wave(B, n16, 71).cuda()
    wave22 = synth.synth_wave(B, n22, 72, sr=22050).cuda()
    lm = 10
    pos = [1, 70, harch.num_frames(n16) - lm]
    tgt = torch.randint(0, 100, (B, lm), generator=torch.Generator().manual_seed(3))
    out = predict_resident(eng, wave, wave22, pos, lm, diagnostics=True, target_labels=tgt)
    # three full passes
    full_inp = eng.vocode(out["mel"], stretch=True)
    full_masked = eng.vocode(out["mel_masked"], stretch=True)
    exp = out["mel_masked"].clone()
    eng.splice_labels(tgt.cuda(), torch.tensor(pos, dtype=torch.int32).cuda(), exp)
    full_exp = eng.vocode(exp, stretch=True)
    torch.cuda.synchronize()
    assert not torch.equal(full_inp, full_masked) and not torch.equal(full_exp, full_masked)
    assert torch.equal(out["hifi_masked"], full_masked)
    assert torch.equal(out["wave"], full_inp), float((out["wave"] - full_inp).abs().max())
    assert torch.equal(out["expected_inpaint"], full_exp), float((out["expected_inpaint"] - full_exp).abs().max())
    # the window really is a fraction of the clip: Lm frames stretch to ~19, + 4 receptive radii of 16 frames
    assert eng.receptive_radius() == 3993 and out["wave"].shape[1] // 256 > 3 * (19 + 4 * 16) - 20


def test_config5_ragged_lengths_are_bucketed_and_match_per_clip_oracle():
    """configs[4]: variable-length clips (here 1.0 / 1.5 / 1.0 / 2.2 s), blind and masked, through predict_ragged:
    every clip must equal the oracle run on that clip alone (mel front-end included)."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.predict import predict_ragged
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    eng, (hsd, gsd, cb) = _mk(harch, varch, K=50)
    secs = [1.0, 1.5, 1.0, 2.2]
    w16 = [synth.synth_wave(1, int(s * 16000), 40 + i)[0].numpy() for i, s in enumerate(secs)]
    w22 = [synth.synth_wave(1, int(s * 16000) * 22050 // 16000, 60 + i, sr=22050)[0].numpy() for i, s in enumerate(secs)]
    pos, lm = [10, 30, 20, 55], 5
    for blind in (False, True):
        outs = predict_ragged(eng, w16, w22, pos, lm, blind=blind, max_batch=2)
        assert len(outs) == 4
        for i in range(4):
            s22, e22 = pos[i] * 320 * 22050 // 16000, (pos[i] + lm) * 320 * 22050 // 16000
            mel = R.masked_mel([w22[i]], None if blind else [s22], None if blind else [e22])
            ref = R.predict_batch(hsd, harch, gsd, varch, cb, torch.from_numpy(w16[i])[None], mel, [pos[i]], lm, blind=blind)
            got = outs[i]
            assert got["wave"].shape == tuple(ref["wave"].shape)
            assert torch.equal(got["labels"].cpu(), ref["labels"]), (blind, i)
            assert rms(got["wave"].cpu(), ref["wave"]) <= 1e-4, (blind, i, rms(got["wave"].cpu(), ref["wave"]))


def test_ida_style_generator_geometry_matches_oracle():
    """SURVEY 8(f) row f-2, generator core: the unit-HiFi-GAN shape of I_da/configs/LJSpeech/hubert_lut.json:13-20 --
    upsample rates (5, 4, 4, 2, 2) with kernels (11, 8, 8, 4, 4) (11 is NOT a multiple of 5: three taps per phase, some
    of them empty), 384 input channels, 16 channels in the last stage -- through si_hifigan_forward(stretch = 0), against
    the oracle's generator (I_da/src/models.py:185-199 is the same forward as I_ea/hifi_gan/models.py:107-123)."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    from speech_inpainting_amd.native import NativeError
    varch = VocoderArch(upsample_rates=(5, 4, 4, 2, 2), upsample_kernel_sizes=(11, 8, 8, 4, 4), upsample_initial_channel=512,
                        num_mels=384, sampling_rate=16000)
    harch = HubertArch.tiny()
    gsd = synth.synth_generator_state(varch)
    eng = InpaintingEngine(harch, varch, 20, "cuda:0", "fp32", "fp32").load_state(synth.synth_hubert_state(harch), gsd, synth.synth_codebook(20))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 384, 23, generator=g) * 0.5                  # 23 frames of 20 ms -> 23 * 320 samples
    taps = {}
    ref = R.generator_forward(gsd, varch, x, taps)[:, 0, :]
    got = eng.vocode(x.cuda(), stretch=False).cpu()
    assert got.shape == ref.shape == (2, 23 * 320)
    err, sig = rms(got, ref), rms(ref, torch.zeros_like(ref))
    print(f"I_da-style generator: waveform rms error {err:.3e} (signal rms {sig:.3f})")
    assert err <= 1e-4 * max(sig, 1.0) and sig > 1e-3
    # the mel-codebook splice does not apply to a generator whose input is not an 80-bin mel
    with pytest.raises((NativeError, AssertionError)):
        eng.splice(torch.zeros(2, 10, 80, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"), 2,
                   torch.zeros(2, 384, 23, device="cuda"))
    # fp16 operand mode on the same shape stays inside the gate
    eng16 = InpaintingEngine(harch, varch, 20, "cuda:0", "fp32", "fp16").load_state(synth.synth_hubert_state(harch), gsd, synth.synth_codebook(20))
    got16 = eng16.vocode(x.cuda(), stretch=False).cpu()
    assert rms(got16, ref) <= 1e-3


def test_ida_code_generator_front_and_decoder_match_oracle():
    """SURVEY 8(f) row f-2, the rest: `CodeGenerator.forward` in the hubert_lut.json shape -- 100 content units and 20 F0
    codes (embedding_dim 128), F0 series at half the unit rate (so `_upsample` repeats every F0 frame twice), one speaker
    embedding per clip, 384 channels into the (5, 4, 4, 2, 2) generator -- against the oracle's restatement of
    I_da/src/model.py:79-119,148-189 (parity unpinned: I_da is not importable) + the oracle's generator."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import CodeGenerator, InpaintingEngine
    from speech_inpainting_amd.native import NativeError
    varch = VocoderArch(upsample_rates=(5, 4, 4, 2, 2), upsample_kernel_sizes=(11, 8, 8, 4, 4), upsample_initial_channel=512,
                        num_mels=384, sampling_rate=16000)
    harch = HubertArch.tiny()
    gsd = synth.synth_generator_state(varch)
    eng = InpaintingEngine(harch, varch, 20, "cuda:0", "fp32", "fp32").load_state(synth.synth_hubert_state(harch), gsd, synth.synth_codebook(20))
    g = torch.Generator().manual_seed(9)
    B, Fc, Fp, E = 2, 28, 14, 128
    emb_c, emb_p = torch.randn(100, E, generator=g) * 0.5, torch.randn(20, E, generator=g) * 0.5
    code = torch.randint(0, 100, (B, Fc), generator=g)
    f0c = torch.randint(0, 20, (B, Fp), generator=g)
    spk = torch.randn(B, E, generator=g) * 0.5
    x_ref = R.code_generator_front(code, emb_c, f0c, emb_p, spk)
    assert x_ref.shape == (B, 384, Fc)
    x = eng.ctx.unit_frontend(code.cuda(), emb_c.cuda(), f0c.cuda(), emb_p.cuda(), spk.cuda()).cpu()
    assert torch.equal(x, x_ref)                                                    # pure data movement: bit-exact
    # F0 at the higher rate instead: the content units are the ones repeated
    x2 = eng.ctx.unit_frontend(code[:, :7].contiguous().cuda(), emb_c.cuda(), f0c.cuda(), emb_p.cuda(), None).cpu()
    assert torch.equal(x2, R.code_generator_front(code[:, :7], emb_c, f0c, emb_p, None))
    # misaligned lengths are refused, as `_upsample` refuses them
    with pytest.raises(NativeError, match="misalignment"):
        eng.ctx.unit_frontend(code[:, :9].contiguous().cuda(), emb_c.cuda(), f0c.cuda(), emb_p.cuda(), None)
    # an index outside the table is visible, not a wild read
    bad = code.clone(); bad[0, 3] = 100
    assert bool(torch.isnan(eng.ctx.unit_frontend(bad.cuda(), emb_c.cuda())[0, :, 3]).all())
    # the whole decoder through the module wrapper
    wav = CodeGenerator(eng, emb_c, emb_p)(code=code, f0_code=f0c, emb=spk).cpu()
    ref = R.generator_forward(gsd, varch, x_ref)
    assert wav.shape == ref.shape == (B, 1, Fc * 320)
    err, sig = rms(wav, ref), rms(ref)
    print(f"CodeGenerator (LUT): waveform rms error {err:.3e} (signal rms {sig:.3f})")
    assert err <= 1e-4 * max(sig, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(3, 800), (1, 64), (2, 1000)])
def test_f0_vqvae_front_matches_oracle(B, T):
    """Row f-2, the F0 side of `CodeGenerator.forward` (I_da/src/model.py:160-166): the fixed VQ-VAE's conv encoder
    (si_f0_encoder_forward: 4 x [stride-2 conv + 4 dilated res blocks] + conv, hubert_lut.json:42-52), its bottleneck's
    arg-min (si_kmeans_assign) and the embedding look-up -- against the oracle's restatement of jukebox.py / resnet.py /
    vq.py (itself pinned by tests/golden/f0_vqvae.npz).  fp32 MACs in a different order: 1e-5 relative; codes equal wherever
    the two nearest bins are not within rounding of each other."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import native, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import F0Quantizer, InpaintingEngine
    harch, varch = HubertArch.tiny(), VocoderArch.v1()
    eng = InpaintingEngine(harch, varch, 20, "cuda:0", "fp32", "fp32").load_state(synth.synth_hubert_state(harch), synth.synth_generator_state(varch),
                                                                                    synth.synth_codebook(20))
    desc = native.F0EncDesc()
    sd = synth.synth_f0_vqvae_state(desc, l_bins=20, seed=5)
    g = torch.Generator().manual_seed(T)
    f0 = torch.randn(B, 1, T, generator=g).abs() * 2.0                 # a normalised, non-negative F0 track
    f0[:, :, T // 3: T // 2] = 0.0                                     # with an unvoiced stretch
    h_ref = R.f0_encoder_forward(sd, f0)                               # (B, 128, T / 16)
    q = F0Quantizer(eng, sd, desc)
    h = q.features(f0.cuda()).cpu()                                    # (B, T / 16, 128) channels-last
    assert h.shape == (B, T // 16, 128) and h_ref.shape == (B, 128, T // 16)
    rel = rms(h, h_ref.transpose(1, 2)) / rms(h_ref)
    print(f"B={B} T={T}: encoder output rms {rms(h_ref):.3f}, relative error {rel:.2e}")
    assert rel <= 1e-5
    z = q(f0.cuda()).cpu()
    z_ref = R.f0_vq_codes(h_ref, sd["vq.level_blocks.0.k"])
    assert z.shape == z_ref.shape == (B, T // 16) and z.dtype == torch.int64
    # where they differ the two bins must be a near-tie for the oracle's own distances
    x = h_ref.permute(0, 2, 1).reshape(-1, 128)
    d = ((x[:, None, :] - sd["vq.level_blocks.0.k"][None]) ** 2).sum(-1)
    dz, dr = d.gather(1, z.reshape(-1, 1)), d.gather(1, z_ref.reshape(-1, 1))
    assert bool(((dz - dr).abs() <= 1e-4 * dr.abs() + 1e-6).all())
    assert (z == z_ref).float().mean().item() >= 0.99
    # through the module wrapper, as the reference calls it: generator(code=..., f0=..., emb=...)
    from speech_inpainting_amd.engine import CodeGenerator
    E = 128
    emb_c, emb_p = torch.randn(100, E, generator=g) * 0.5, torch.randn(20, E, generator=g) * 0.5
    code = torch.randint(0, 100, (B, T // 16), generator=g)
    gen = CodeGenerator(eng, emb_c, emb_p, f0_quantizer=q)
    x_a = eng.ctx.unit_frontend(code.cuda(), gen.emb_c, z.cuda(), gen.emb_p, None)
    x_b = eng.ctx.unit_frontend(code.cuda(), gen.emb_c, q(f0.cuda()), gen.emb_p, None)
    assert torch.equal(x_a, x_b) and x_a.shape == (B, 2 * E, T // 16)
    # wrong weight count / too few frames are refused
    with pytest.raises(ValueError):
        eng.ctx.f0_encoder(desc, q.weights[:-1].contiguous(), f0.cuda())
    with pytest.raises(ValueError):
        eng.ctx.f0_encoder(desc, q.weights, f0[:, :, :8].contiguous().cuda())


def test_f0_encoder_single_launch_equals_the_layer_by_layer_form():
    """si_f0_encoder_forward as ONE persistent launch (activations in LDS) against its 37-launch form (SI_F0_FUSED=0, also the route
    of tracks too long for LDS): bit-identical, for the hubert_lut.json shape at 4 s / 10 s tracks and a 30 s track that does not fit."""
    import os
    from speech_inpainting_amd import native, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    ctx = native.NativeContext(native.make_desc(HubertArch.tiny(), VocoderArch.tiny(), 10), torch.device("cuda:0"))
    desc = native.F0EncDesc()
    from speech_inpainting_amd.engine import pack_f0_encoder
    w = pack_f0_encoder(synth.synth_f0_vqvae_state(desc, 20, seed=5), desc).cuda()
    g = torch.Generator().manual_seed(9)
    for B, T in ((16, 797), (3, 2000), (2, 6000)):
        f0 = torch.randn(B, 1, T, generator=g).cuda()
        a = ctx.f0_encoder(desc, w, f0)
        os.environ["SI_F0_FUSED"] = "0"
        try:
            b = ctx.f0_encoder(desc, w, f0)
        finally:
            os.environ.pop("SI_F0_FUSED", None)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (B, T, float((a - b).abs().max()))
    ctx.close()


def test_f0_vqvae_front_matches_reference_goldens():
    """Row f-2 against the REFERENCE: `si_f0_encoder_forward` + `si_kmeans_assign` on the inputs of tests/golden/f0_vqvae.npz
    against the outputs of the reference's own `Encoder` / `Bottleneck` modules (I_da/src/modules/jukebox.py, vq.py; fixture
    written by tools/make_goldens.py::f0_vqvae_cases): encoder output <= 1e-5 relative, codes identical except where the
    reference's own two best distances are within fp32 rounding of each other."""
    import os
    from speech_inpainting_amd import native, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import F0Quantizer, InpaintingEngine
    from tests.common import GOLDEN
    z = np.load(os.path.join(GOLDEN, "f0_vqvae.npz"))
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    eng = InpaintingEngine(harch, varch, 20, "cuda:0", "fp32", "fp32")
    desc = native.F0EncDesc()
    sd = synth.synth_f0_vqvae_state(desc, 20, seed=11)
    q = F0Quantizer(eng, sd, desc)
    k = sd["vq.level_blocks.0.k"]
    for T in (64, 800, 1000):
        f0 = torch.from_numpy(z[f"f0_{T}"])
        ref = torch.from_numpy(z[f"h_{T}"])                                       # (B, 128, T / 16)
        h = q.features(f0.cuda()).cpu()                                           # (B, T / 16, 128)
        rel = rms(h, ref.transpose(1, 2)) / rms(ref)
        codes = q(f0.cuda()).cpu()
        want = torch.from_numpy(z[f"codes_{T}"])
        agree = float((codes == want).float().mean())
        print(f"T={T}: encoder output vs reference {rel:.2e} relative, codes agree {agree:.3f}")
        assert rel <= 1e-5
        x = ref.permute(0, 2, 1).reshape(-1, 128)
        d = ((x[:, None, :] - k[None]) ** 2).sum(-1)
        dz, dr = d.gather(1, codes.reshape(-1, 1)), d.gather(1, want.reshape(-1, 1))
        assert bool(((dz - dr).abs() <= 1e-4 * dr.abs() + 1e-6).all()) and agree >= 0.98


@pytest.mark.gpu
def test_local_huggingface_directory_loads_and_encodes_like_the_pt_route(tmp_path):
    """The "HuggingFace checkpoint loader": a LOCAL directory (config.json + model.safetensors, `hubert.`-prefixed keys as in a
    HubertForCTC file) supplies the architecture and the encoder weights in place of `from_pretrained(name)` (I_ea/model.py:26-40);
    with the head taken from the CustomModel .pt it must encode exactly like the .pt route (I_ea/predict.py:149)."""
    import json
    from safetensors.torch import save_file
    from speech_inpainting_amd import checkpoint, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(100)
    d = tmp_path / "hubert-tiny"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(harch.to_hf()))
    save_file({"hubert." + k[len("base_model."):]: v.contiguous() for k, v in hsd.items() if k.startswith("base_model.")}, str(d / "model.safetensors"))
    torch.save(hsd, tmp_path / "save_checkpoint.pt")
    sd_dir, arch_dir = checkpoint.load_hubert_checkpoint(str(d))
    sd_pt, _ = checkpoint.load_hubert_checkpoint(str(tmp_path / "save_checkpoint.pt"), "base")
    assert arch_dir == harch and not any(k.startswith("final_layers") for k in sd_dir)
    sd_dir.update({k: v for k, v in sd_pt.items() if k.startswith("final_layers.")})
    wave = synth.synth_wave(2, 8000, 5).cuda()
    a = InpaintingEngine(arch_dir, varch, 100, "cuda:0", "bf16", "fp16").load_state(sd_dir, gsd, cb).encode(wave)
    b = InpaintingEngine(harch, varch, 100, "cuda:0", "bf16", "fp16").load_state(sd_pt, gsd, cb).encode(wave)
    assert torch.equal(a, b)
    # the directory alone (no trained head, no codebook): the encoder-only entry point serves, the I_ea calls that need the
    # missing parts refuse instead of computing on placeholders
    e = InpaintingEngine(arch_dir, varch, 100, "cuda:0").load_state(checkpoint.load_hubert_checkpoint(str(d))[0], gsd, None)
    hid = e.extract_features(wave, harch.num_hidden_layers)
    assert hid.shape == (2, harch.num_frames(8000), harch.hidden_size) and bool(torch.isfinite(hid).all())
    with pytest.raises(RuntimeError, match="final_layers"):
        e.encode(wave)
    with pytest.raises(RuntimeError, match="codebook"):
        e.splice(a, torch.zeros(2, dtype=torch.int32, device="cuda"), 2, torch.zeros(2, 80, 30, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("voc,tol", [("fp32", 1e-6), ("bf16x3", 1e-5), ("fp16", 2e-4), ("bf16", 1e-3)])
def test_resblock2_generator_matches_reference_golden(voc, tol):
    """config_v3.json generators (`resblock: "2"`, I_ea/hifi_gan/models.py:52-73): one dilated conv per dilation with the residual, the
    1 / num_kernels scale and the MRF accumulate in its epilogue -- against the output of the reference's own `Generator(h)`
    (tests/golden/gen_v3.npz) in every arithmetic mode."""
    import os
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import Generator, InpaintingEngine
    from tests.common import GOLDEN
    z = np.load(os.path.join(GOLDEN, "gen_v3.npz"))
    harch, varch = HubertArch.tiny(), VocoderArch.v3()
    gsd = synth.synth_generator_state(varch, synth.DEFAULT_SEED + 1)
    mel = synth.synth_mel(2, 40, 80, synth.DEFAULT_SEED + 4)
    eng = InpaintingEngine(harch, varch, 100, "cuda:0", "fp32", voc).load_state(synth.synth_hubert_state(harch), gsd, synth.synth_codebook(100))
    wav = Generator(eng)(mel.cuda())[:, 0, :].cpu()
    ref = torch.from_numpy(z["wave"])
    err = rms(wav, ref)
    print(f"ResBlock2 generator, vocoder {voc}: waveform rms error {err:.3e} (signal rms {float(z['wave_rms']):.3f})")
    assert wav.shape == ref.shape and err <= tol
    # folded weights (after remove_weight_norm) load to the same result
    if voc == "fp32":
        eng2 = InpaintingEngine(harch, varch, 100, "cuda:0").load_state(synth.synth_hubert_state(harch), synth.synth_generator_state(varch, synth.DEFAULT_SEED + 1, folded=True),
                                                                       synth.synth_codebook(100))
        assert rms(Generator(eng2)(mel.cuda())[:, 0, :].cpu(), ref) <= tol
