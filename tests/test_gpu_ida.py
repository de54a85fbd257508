"""SURVEY 8(f) row f-2 as a PATH: I_da's `inpainting()` (I_da/scripts/inpainting.py:151-266) through the C ABI --
si_hubert_extract_features (encoder output at `output_layer` with I_da's corruption and whole-clip layer norm fused into the first
conv), si_kmeans_assign, si_code_splice, si_f0_encoder_forward, si_unit_frontend, si_hifigan_forward -- against the reference
goldens where the reference can produce them and against the CPU oracle's restatement of the script."""
import json
import os

import numpy as np
import pytest
import torch

from tests.common import GOLDEN, rms

pytestmark = pytest.mark.gpu

UNIT_VARCH = dict(upsample_rates=(5, 4, 4, 2, 2), upsample_kernel_sizes=(11, 8, 8, 4, 4), upsample_initial_channel=512,
                  num_mels=384, sampling_rate=16000)                      # I_da/configs/LJSpeech/hubert_lut.json:13-20,66


def _layer_arch(**kw):
    from speech_inpainting_amd.arch import HubertArch
    return HubertArch.tiny(conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True, **kw)


@pytest.mark.parametrize("tag", ["group", "layer"])
def test_extract_features_matches_hidden_state_goldens(tag):
    """`si_hubert_extract_features` (fp32) against `HubertModel(..., output_hidden_states=True).hidden_states[L]` on inputs prepared
    by the reference's own statements (`(y + 1e-6) * mask` in float64, `.float()`, `F.layer_norm(x, x.shape)`), both encoder
    flavours, every stored layer, clean and corrupted clips (tests/golden/hidden_layers.npz)."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    from speech_inpainting_amd.native import NativeError
    z = np.load(os.path.join(GOLDEN, "hidden_layers.npz"))
    meta = json.loads(str(z["meta"]))
    harch = HubertArch.tiny(num_hidden_layers=3) if tag == "group" else _layer_arch(num_hidden_layers=3)
    hsd = synth.synth_hubert_state(harch, meta["seed"] + 60)
    y = synth.synth_wave(2, meta["N"], meta["seed"] + 61)
    assert np.allclose([float(hsd["base_model.encoder.layers.2.attention.q_proj.weight"][0, 0]), float(y[1][100])], z[f"{tag}_probe"], atol=1e-7)
    # the encoder alone: no head in the state dict (load_state initialises one), no mel codebook
    enc_only = {k: v for k, v in hsd.items() if k.startswith("base_model.")}
    varch = VocoderArch.tiny()
    eng = InpaintingEngine(harch, varch, 100, "cuda:0", "fp32", "fp32").load_state(enc_only, synth.synth_generator_state(varch))
    wave = torch.cat([y, y]).cuda()                                             # clean 0, 1 then corrupted 0, 1
    ms = torch.tensor([0, 0, meta["frame_start"], meta["frame_start"]], dtype=torch.int32).cuda()
    ml = torch.tensor([0, 0, meta["mask_size"], meta["mask_size"]], dtype=torch.int32).cuda()
    add = torch.tensor([0.0, 0.0, 1e-6, 1e-6], dtype=torch.float64).cuda()
    for L in ((1, 2, 3) if tag == "group" else (1, 2)):
        h = eng.extract_features(wave, L, "layer_norm", ms, ml, add).cpu()
        for i, (kind, b) in enumerate((("clean", 0), ("clean", 1), ("masked", 0), ("masked", 1))):
            ref = torch.from_numpy(z[f"{tag}_{kind}_{b}_L{L}"])
            rel = rms(h[i], ref) / rms(ref)
            print(f"{tag} L={L} {kind} clip {b}: {rel:.2e} relative")
            assert h[i].shape == ref.shape and rel <= 2e-5
        # the corruption matters: the corrupted clip's features differ from the clean clip's
        assert rms(h[2], h[0]) > 1e-2 * rms(h[0])
    with pytest.raises(NativeError, match="output_layer"):
        eng.extract_features(wave, 4)
    with pytest.raises(NativeError, match="output_layer"):
        eng.extract_features(wave, 0)
    # the same call without I_da's prologue pieces is the I_ea encoder's hidden state: `last_hidden` of si_hubert_forward
    if tag == "group":
        cap = eng.ctx.capture(["last_hidden"], capacity=2 * 24 * harch.hidden_size)
        eng.ctx.hubert_forward(y.cuda(), None, None, True)      # (native call: only the hidden state in front of the absent head is read)
        h3 = eng.extract_features(y.cuda(), 3, "processor").cpu()
        torch.cuda.synchronize()
        assert torch.equal(h3.reshape(-1), cap["last_hidden"].cpu())
        eng.ctx.clear_captures()


def test_code_splice_is_the_scripts_slice_assignment():
    from oracle import ref_cpu as R
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    eng = InpaintingEngine(HubertArch.tiny(), VocoderArch.tiny(), 100, "cuda:0")
    g = torch.Generator().manual_seed(3)
    B, T, hop = 7, 199, 320
    code = torch.randint(0, 100, (B, T), generator=g)
    pred = torch.randint(0, 100, (B, T), generator=g)
    starts = [24000, 0, 63000, 320 * 198, 100, 64000, 319]                # inside, at 0, past the end, last frame, odd offsets
    ms = 6400
    first = torch.tensor([s // hop for s in starts], dtype=torch.int32)
    last = torch.tensor([(s + ms) // hop for s in starts], dtype=torch.int32)
    out = eng.ctx.code_splice(code.cuda(), pred.cuda(), first.cuda(), last.cuda()).cpu()
    for b in range(B):
        assert torch.equal(out[b], R.code_splice(code[b], pred[b], starts[b], ms, hop)), b
    # an empty / inverted span keeps the clean series
    out2 = eng.ctx.code_splice(code.cuda(), pred.cuda(), last.cuda(), first.cuda()).cpu()
    assert torch.equal(out2, code)


def _ida_setup(harch, B, N, seed, K=100):
    from speech_inpainting_amd import native, synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch(**UNIT_VARCH)
    hsd = {k: v for k, v in synth.synth_hubert_state(harch, seed).items() if k.startswith("base_model.")}
    gsd = synth.synth_generator_state(varch, seed + 1)
    f0sd = synth.synth_f0_vqvae_state(native.F0EncDesc(), 20, seed=seed + 2)
    g = torch.Generator().manual_seed(seed + 3)
    E = 128
    emb_c, emb_p = torch.randn(K, E, generator=g) * 0.5, torch.randn(20, E, generator=g) * 0.5
    spk = torch.randn(B, E, generator=g) * 0.5
    wave = synth.synth_wave(B, N, seed + 4)
    Tf0 = N // 80 - 3                                                          # a YAAPT-like frame count (5 ms hop)
    t = torch.arange(Tf0, dtype=torch.float32)
    f0 = torch.stack([(torch.sin(t / (23.0 + b)) + 0.3 * torch.randn(Tf0, generator=g)) * (torch.sin(t / 61.0 + b) > -0.4).float() for b in range(B)])[:, None, :]
    return varch, hsd, gsd, f0sd, emb_c, emb_p, spk, wave, f0


@pytest.mark.parametrize("name,B,N,L,K", [("large", 16, 64000, 18, 100), ("tiny_layer", 3, 9600, 2, 50), ("tiny_group", 2, 12800, 1, 500)])
def test_ida_inpainting_path_matches_oracle(name, B, N, L, K):
    """configs[3] as the reference runs it: HuBERT-large (24 pre-LN layers, LayerNorm feature extractor) at `output_layer`, 400 ms
    mask, 16 clips (the per-GPU share of batch 128 on 8 GPUs), k-means units, unit splice, F0 VQ-VAE, CodeGenerator -- the whole
    `inpainting()` on the GPU in fp32 against the oracle's clip-by-clip restatement: features <= 1e-4 relative, units identical
    except at near-ties of the oracle's own distances, both waveforms <= 1e-4 RMS."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd.arch import HubertArch
    from speech_inpainting_amd.engine import CodeGenerator, F0Quantizer, InpaintingEngine
    harch = {"large": HubertArch.large(), "tiny_layer": _layer_arch(), "tiny_group": HubertArch.tiny()}[name]
    varch, hsd, gsd, f0sd, emb_c, emb_p, spk, wave, f0 = _ida_setup(harch, B, N, 77, K)
    frame_start, mask_size = (24000, 6400) if name == "large" else (3200, 1600)      # 1.5 s / 400 ms (inpainting.py:188, main :345-360)
    torch.set_num_threads(16)
    # centroids: noisy copies of real feature rows, so the assignment has structure (several units, real decisions)
    with torch.no_grad():
        f_ref0 = R.hubert_get_feats(hsd, harch, wave[0].numpy().astype(np.float64), L)
    g = torch.Generator().manual_seed(5)
    rows = torch.randint(0, f_ref0.shape[0], (K,), generator=g)
    cent = f_ref0[rows] + 0.25 * f_ref0.std() * torch.randn(K, f_ref0.shape[1], generator=g)
    eng = InpaintingEngine(harch, varch, K, "cuda:0", "fp32", "fp32").load_state(hsd, gsd)
    gen = CodeGenerator(eng, emb_c, emb_p, f0_quantizer=F0Quantizer(eng, f0sd))
    out = eng.ida_inpaint_batch(wave.cuda(), frame_start, mask_size, cent, gen, f0, spk, output_layer=L)
    torch.cuda.synchronize()
    T = harch.num_frames(N)
    worst_f = worst_w = 0.0
    differing = 0
    for b in range(B):
        ref = R.ida_inpaint(hsd, harch, gsd, varch, cent, emb_c, emb_p, f0sd, wave[b].numpy().astype(np.float64), frame_start, mask_size,
                            f0[b], spk[b], output_layer=L)
        for i, key in ((b, "feats"), (B + b, "feats_inpainting")):
            worst_f = max(worst_f, rms(out["feats"][i].cpu(), ref[key]) / rms(ref[key]))
        for key in ("code", "code_inpainting"):
            got, want = out[key][b].cpu(), ref[key]
            assert got.shape == want.shape
            if not torch.equal(got, want):
                # a differing unit must be a near-tie of the oracle's own distances
                feats = ref["feats" if key == "code" else "feats_inpainting"][: got.numel()]
                d = ((feats[:, None, :] - cent[None]) ** 2).sum(-1)
                dg, dw = d.gather(1, got.reshape(-1, 1)), d.gather(1, want.reshape(-1, 1))
                assert bool(((dg - dw).abs() <= 1e-4 * dw.abs() + 1e-5).all()), (b, key)
                differing += int((got != want).sum())
        if torch.equal(out["code"][b].cpu(), ref["code"]) and torch.equal(out["code_inpainting"][b].cpu(), ref["code_inpainting"]):
            for key in ("audio_gen", "audio_inp"):
                assert out[key][b].shape == ref[key].shape
                worst_w = max(worst_w, rms(out[key][b].cpu(), ref[key]))
        if b == 0:
            # the splice did something: inside the mask the corrupted clip's own units survive
            lo, hi = frame_start // 320, (frame_start + mask_size) // 320
            assert torch.equal(ref["code_inpainting"][:lo], ref["code"][:lo]) and len(set(ref["code"].tolist())) >= 5
            print(f"{name}: units in the mask  clean {ref['code'][lo:hi].tolist()}\n{' ' * len(name)}            corrupted {ref['code_inpainting'][lo:hi].tolist()}")
    n_units = 2 * B * out["code"].shape[1]
    print(f"{name}: B={B} T={T} L={L} K={K}: features {worst_f:.2e} relative, {differing}/{n_units} units differ (near-ties), waveforms {worst_w:.2e} RMS "
          f"(signal {rms(out['audio_inp'].cpu()):.3f}), output {tuple(out['audio_inp'].shape)}")
    assert worst_f <= 1e-4 and differing <= max(1, n_units // 500) and worst_w <= 1e-4
    a, nc, nci, nf = R.ida_match_lengths(N, T, f0.shape[-1])
    assert out["audio_gen"].shape == (B, nc * 320) and out["audio_inp"].shape == (B, nci * 320)


def test_ida_path_in_the_timed_arithmetic_units_are_near_ties_and_waveforms_meet_the_gate():
    """The arithmetic `bench.py` TIMES for configs[3] (`configs3_ida`: bf16 encoder + fp16 unit vocoder, the 16 -> 32-channel padded
    last stage) against the fp32 oracle, at configs[3]'s per-GPU shape (HuBERT-large, layer 18, 16 clips x 4 s, 400 ms mask, K = 100).
    The k-means arg-min over 1024-dim features is a discrete decision (like A12): (1) unit agreement >= the measured floor;
    (2) every unit that differs is a near-tie of the ORACLE's own distances -- for x' = x + e the winner g of x' and the winner w of x
    satisfy d(x, c_g) - d(x, c_w) <= 2 |e| |c_g - c_w| exactly (expand |x + e - c|^2), with e this run's measured feature error;
    (3) both waveforms within 1e-3 RMS of the oracle's CodeGenerator (front + F0 VQ-VAE + unit HiFi-GAN, fp32) on THIS run's units."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd.arch import HubertArch
    from speech_inpainting_amd.engine import CodeGenerator, F0Quantizer, InpaintingEngine
    harch = HubertArch.large()
    B, N, L, K = 16, 64000, 18, 100
    varch, hsd, gsd, f0sd, emb_c, emb_p, spk, wave, f0 = _ida_setup(harch, B, N, 77, K)
    frame_start, mask_size = 24000, 6400
    torch.set_num_threads(16)
    with torch.no_grad():
        f_ref0 = R.hubert_get_feats(hsd, harch, wave[0].numpy().astype(np.float64), L)
    g = torch.Generator().manual_seed(5)
    rows = torch.randint(0, f_ref0.shape[0], (K,), generator=g)
    cent = f_ref0[rows] + 0.25 * f_ref0.std() * torch.randn(K, f_ref0.shape[1], generator=g)
    eng = InpaintingEngine(harch, varch, K, "cuda:0", "bf16", "fp16").load_state(hsd, gsd)
    gen = CodeGenerator(eng, emb_c, emb_p, f0_quantizer=F0Quantizer(eng, f0sd))
    out = eng.ida_inpaint_batch(wave.cuda(), frame_start, mask_size, cent, gen, f0, spk, output_layer=L)
    torch.cuda.synchronize()
    T = harch.num_frames(N)
    units_run = torch.cat([eng.ctx.kmeans_assign(out["feats"].reshape(-1, harch.hidden_size), cent.cuda().contiguous()).reshape(2 * B, T).cpu()])
    n_units = n_diff = 0
    worst_rel = worst_w = 0.0
    sample = (0, 5, 11, 15)                                          # the oracle's encoder costs ~2 s per clip and stream
    for b in sample:
        y = wave[b].numpy().astype(np.float64)
        with torch.no_grad():
            refs = (R.hubert_get_feats(hsd, harch, y, L), R.hubert_get_feats(hsd, harch, R.ida_corrupt(y, frame_start, mask_size), L))
        for i, x in ((b, refs[0]), (B + b, refs[1])):
            xr = out["feats"][i].cpu()
            worst_rel = max(worst_rel, rms(xr, x) / rms(x))
            want = R.kmeans_assign(x, cent)
            got = units_run[i]
            d = ((x[:, None, :] - cent[None]) ** 2).sum(-1)
            e = (xr - x).norm(dim=1)
            for t in (want != got).nonzero().reshape(-1).tolist():
                margin = float(d[t, got[t]] - d[t, want[t]])
                bound = 2.0 * float(e[t]) * float((cent[got[t]] - cent[want[t]]).norm())
                assert -1e-3 <= margin <= bound * (1 + 1e-4) + 1e-3, (b, i, t, margin, bound)
            n_units += want.numel()
            n_diff += int((want != got).sum())
        # (3) the waveforms against the oracle's generator on THIS run's units
        with torch.no_grad():
            _, nc, nci, nf = R.ida_match_lengths(N, T, f0.shape[-1])
            z_p = R.f0_vq_codes(R.f0_encoder_forward(f0sd, f0[b][None, :, :nf].float()), f0sd["vq.level_blocks.0.k"])
            for key, code in (("audio_gen", out["code"][b].cpu()), ("audio_inp", out["code_inpainting"][b].cpu())):
                ref = R.generator_forward(gsd, varch, R.code_generator_front(code[None], emb_c, z_p, emb_p, spk[b][None]))[0, 0]
                assert out[key][b].shape == ref.shape
                worst_w = max(worst_w, rms(out[key][b].cpu(), ref))
    agree = 1.0 - n_diff / n_units
    print(f"I_da in bf16 / fp16: feature error {worst_rel:.3e} relative, unit agreement {agree:.4f} ({n_diff} of {n_units} differ, all near-ties within "
          f"2 |e| |c_g - c_w|), waveforms vs the oracle's generator on this run's units {worst_w:.3e} RMS (signal {rms(out['audio_inp'].cpu()):.3f})")
    assert worst_rel <= 3e-2 and agree >= 0.95 and worst_w <= 1e-3 and worst_w <= 5e-3 * rms(out['audio_inp'].cpu())
    # the run's own splice: inside the mask the corrupted stream's units, outside the clean stream's
    lo, hi = frame_start // 320, (frame_start + mask_size) // 320
    nc = out["code"].shape[1]
    assert torch.equal(out["code_inpainting"][:, :lo].cpu(), units_run[:B, :lo]) and torch.equal(out["code_inpainting"][:, lo:hi].cpu(), units_run[B:, lo:hi])
    assert torch.equal(out["code"].cpu(), units_run[:B, :nc])
