"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures."""
import numpy as np
import pytest
import torch

from tests.common import load_case, rms

pytestmark = pytest.mark.gpu


def _engine(c, enc="fp32", voc="fp32", chunk=0):
    from speech_inpainting_amd.engine import InpaintingEngine
    eng = InpaintingEngine(c["harch"], c["varch"], c["meta"]["K"], "cuda:0", enc, voc, chunk)
    return eng.load_state(c["hsd"], c["gsd"], c["cb"])


def _run(eng, c):
    m = c["meta"]
    pos = torch.tensor(c["frame_pos"], dtype=torch.int32, device="cuda")
    out = eng.predict_batch(c["wave"].cuda(), c["mel"].cuda(), pos, m["lm"], blind=m["blind"])
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}


@pytest.mark.parametrize("name", ["tiny_group", "tiny_layer", "tiny_blind", "base_4s", "large_4s", "base_b4"])
def test_fp32_matches_reference_goldens(name):
    """fp32 mode vs outputs of the reference's own modules: labels bit-exact, waveform RMS <= 1e-3 (north-star gate)."""
    c = load_case(name)
    z = c["z"]
    out = _run(_engine(c), c)
    feats_ref = torch.from_numpy(z["feats"])
    assert out["feats"].shape == feats_ref.shape
    assert rms(out["feats"], feats_ref) <= 1e-4 * max(rms(feats_ref), 1.0)
    assert np.array_equal(out["labels"].numpy(), z["labels"])
    assert np.allclose(out["mel"].numpy(), z["mel_spliced"], rtol=0, atol=1e-6)
    if "wave" in z.files:
        err = rms(out["wave"], z["wave"])
    else:
        err = max(rms(out["wave"][:, :2048], z["wave_head"]), rms(out["wave"][:, -2048:], z["wave_tail"]))
    if "wave_win" in z.files:         # base_b4: the samples each clip's spliced frames reach
        err = max([err] + [rms(out["wave"][i, lo:lo + 16384], z["wave_win"][i]) for i, lo in enumerate(z["wave_win_lo"])])
    print(f"{name}: waveform rms error {err:.3e} (signal rms {float(z['wave_rms']):.3f})")
    assert err <= 1e-3
    assert err <= 1e-4            # what exact-fp32 MFMA actually delivers; keeps regressions visible


@pytest.mark.parametrize("name", ["tiny_group", "tiny_layer"])
def test_stage_taps_match_oracle(name):
    """Per-stage intermediates (si_debug_tensor) vs the oracle: localises a wrong kernel."""
    from oracle import ref_cpu as R
    c = load_case(name)
    m = c["meta"]
    taps = {}
    R.predict_batch(c["hsd"], c["harch"], c["gsd"], c["varch"], c["cb"], c["wave"], c["mel"], c["frame_pos"], m["lm"],
                    blind=m["blind"], taps=taps)
    eng = _engine(c, chunk=8)
    pos = torch.tensor(c["frame_pos"], dtype=torch.int32, device="cuda")
    sl = [R.mask_samples_from_frames(p, m["lm"]) for p in c["frame_pos"]]
    ms = torch.tensor([s for s, _ in sl], dtype=torch.int32, device="cuda")
    ml = torch.tensor([l for _, l in sl], dtype=torch.int32, device="cuda")
    enc_names = ("features", "projected", "encoder_in", "last_hidden")
    voc_names = [f"{k}{i}" for i in range(4) for k in ("ups", "stage")]
    wave, mel0 = c["wave"].cuda(), c["mel"].cuda()
    feats = eng.encode(wave, ms, ml)                       # first pass records the sizes
    mel2 = mel0.clone()
    eng.splice(feats, pos, m["lm"], mel2)
    eng.vocode(mel2)
    caps = eng.ctx.capture(enc_names + tuple(voc_names))
    eng.encode(wave, ms, ml)
    eng.vocode(mel2)
    torch.cuda.synchronize()
    eng.ctx.clear_captures()
    for nm in enc_names:
        got, ref = caps[nm].cpu(), taps[nm].reshape(-1)
        assert got.numel() == ref.numel(), nm
        assert rms(got, ref) <= 2e-5 * max(rms(ref), 1.0), nm
    for nm in voc_names:
        got = caps[nm].cpu()
        ref = taps[nm].transpose(1, 2).reshape(-1)         # oracle is channels-first, the library channels-last
        assert got.numel() == ref.numel(), nm
        assert rms(got, ref) <= 2e-5 * max(rms(ref), 1.0), nm


def _masked_wave_span(c, margin_frames=16):
    """Waveform samples that can depend on the spliced mel frames: the masked frames after the x441/256 stretch, widened
    by the generator's receptive field (conv_pre 3 frames + MRF halos 60 rows per stage = 1920 + 240 + 120 + 60 samples
    + upsamplers: < 14 stretched frames per side)."""
    m = c["meta"]
    lo = min(c["frame_pos"]) * 441 // 256 - margin_frames
    hi = (max(c["frame_pos"]) + m["lm"]) * 441 // 256 + 1 + margin_frames
    return max(lo, 0) * 256, hi * 256


# measured on MI355X (round 2): feats relative rms error and label agreement of the bf16 encoder against the reference's
# fp32 goldens; the asserted bounds are 2x the measured error / the measured agreement floor
# (measured: feats rel 9.0e-3 / 6.8e-3 / 7.4e-3 / 9.3e-3; agreement 10/10, 20/20, 10/10, 28/30)
BF16_FEAT_REL = {"base_4s": 2e-2, "large_4s": 1.5e-2, "tiny_layer": 1.5e-2, "tiny_group": 2e-2}
BF16_AGREE_FLOOR = {"base_4s": 0.9, "large_4s": 0.9, "tiny_layer": 0.9, "tiny_group": 0.85}


def _check_wave(name, c, out, gate):
    """Waveform vs the golden: whole clip when every label agrees, otherwise everything OUTSIDE the receptive field of
    the spliced frames (a flipped codeword legitimately changes the samples it reaches)."""
    z = c["z"]
    agree = float((out["labels"].numpy() == z["labels"]).mean())
    lo, hi = _masked_wave_span(c)
    if "wave" in z.files:
        ref, got = torch.from_numpy(z["wave"]), out["wave"]
        if agree == 1.0:
            err, where = rms(got, ref), "whole clip"
        else:
            err = max(rms(got[:, :lo], ref[:, :lo]) if lo > 0 else 0.0, rms(got[:, hi:], ref[:, hi:]) if hi < ref.shape[1] else 0.0)
            where = f"outside samples [{lo}, {hi})"
    else:     # large_4s stores the first / last 2048 samples, both outside the masked span's receptive field
        assert lo >= 2048 and hi <= out["wave"].shape[1] - 2048
        err, where = max(rms(out["wave"][:, :2048], z["wave_head"]), rms(out["wave"][:, -2048:], z["wave_tail"])), "head + tail"
    print(f"{name}: label agreement {agree:.2f}, waveform rms error {err:.3e} ({where}; signal rms {float(z['wave_rms']):.3f})")
    assert err <= gate, (name, err, where)
    if agree < 1.0:
        # the samples a flipped codeword reaches: the oracle's vocoder on the mel THIS run spliced must give this waveform
        from oracle import ref_cpu as R
        ref2 = R.generator_forward(c["gsd"], c["varch"], R.extend_mel(out["mel"]))[:, 0, :]
        err2 = rms(out["wave"], ref2)
        print(f"{name}: vs the oracle's vocoder on the spliced mel of this run: waveform rms error {err2:.3e}")
        assert err2 <= gate, (name, err2)
    return agree, err


@pytest.mark.parametrize("name", ["base_4s", "large_4s", "tiny_layer", "tiny_group"])
def test_bf16_encoder_mode_label_agreement_and_waveform(name):
    """BASELINE config #2 encoder arithmetic: bf16 MFMA encoder (fp32 accumulate, fp32 head) with operand-ready bf16
    activations, on the post-LN / group-norm flavour (base, tiny_group) and the pre-LN / layer-norm flavour (large,
    tiny_layer), fp32 vocoder.  The arg-max in the middle of the path is a discrete decision: agreement has a stated
    floor, and the waveform is held to the 1e-3 gate wherever the labels cannot reach it."""
    c = load_case(name)
    z = c["z"]
    out = _run(_engine(c, enc="bf16"), c)
    feats_ref = torch.from_numpy(z["feats"])
    rel = rms(out["feats"], feats_ref) / rms(feats_ref)
    print(f"{name} bf16 encoder: feats relative rms error {rel:.3e}")
    assert rel <= BF16_FEAT_REL[name]
    agree, _ = _check_wave(name + " bf16/fp32", c, out, 1e-3)
    assert agree >= BF16_AGREE_FLOOR[name]


def test_bf16_labels_flip_only_where_the_fp32_margin_is_inside_the_feature_error():
    """The base-size fixture with DISCRIMINATING labels (base_b4: 4 clips, different mask positions, a head centred on the masked
    frames -> 8 distinct codewords over 40 frames, several inside one clip) under the bench's encoder arithmetic.  With a centred
    head the decisions hang on the frame-to-frame deviations of a randomly initialised encoder (2 % of the feature norm), the
    same size as bf16's feature error, so some labels flip -- as they must.  What is asserted: (1) every flipped label is a
    near-tie for the REFERENCE's own cosines: its fp32 margin is no larger than twice the distance between the unit vectors of
    the bf16 and the reference feature (no cosine can move further than that); (2) the agreement stays above the measured
    floor; (3) the waveform meets the gate wherever the agreeing labels reach, and equals the oracle's vocoder on the mel this
    run spliced everywhere else."""
    import torch.nn.functional as F
    from oracle import ref_cpu as R
    c = load_case("base_b4")
    z, m = c["z"], c["meta"]
    out = _run(_engine(c, enc="bf16", voc="fp16"), c)
    ref_f = torch.from_numpy(z["feats"])
    lm = m["lm"]
    v_ref = torch.stack([ref_f[b, p:p + lm] for b, p in enumerate(c["frame_pos"])]).reshape(-1, 80)
    v_got = torch.stack([out["feats"][b, p:p + lm] for b, p in enumerate(c["frame_pos"])]).reshape(-1, 80)
    _, cc = R.codebook_tables(c["cb"])
    sim = F.cosine_similarity(v_ref[:, None, :], cc[None], dim=-1)                   # the reference's own cosines
    want, got = torch.from_numpy(z["labels"]).reshape(-1), out["labels"].reshape(-1)
    du = (F.normalize(v_ref, dim=1) - F.normalize(v_got, dim=1)).norm(dim=1)         # how far each unit feature moved
    flipped = (want != got).nonzero().reshape(-1).tolist()
    agree = 1.0 - len(flipped) / want.numel()
    print(f"base_b4 bf16 encoder: agreement {agree:.3f} ({len(flipped)} of {want.numel()} flipped), unit-feature shift median {float(du.median()):.3e} "
          f"max {float(du.max()):.3e}, distinct reference labels {len(set(want.tolist()))}")
    for i in flipped:
        margin = float(sim[i, want[i]] - sim[i, got[i]])
        print(f"    frame {i}: reference label {int(want[i])} -> {int(got[i])}, fp32 margin {margin:.3e}, bound {2 * float(du[i]):.3e}")
        assert 0.0 <= margin <= 2.0 * float(du[i]) + 1e-6
    assert agree >= 0.6
    assert bool(torch.isfinite(out["wave"]).all())
    # the waveform: the oracle's vocoder on the mel THIS run spliced (flipped codewords legitimately change what they reach)
    ref2 = R.generator_forward(c["gsd"], c["varch"], R.extend_mel(out["mel"]))[:, 0, :]
    err2 = rms(out["wave"], ref2)
    head = max(rms(out["wave"][:, :2048], z["wave_head"]), rms(out["wave"][:, -2048:], z["wave_tail"]))
    print(f"base_b4 headline arithmetic: waveform vs the oracle's vocoder on this run's mel {err2:.3e}; head / tail vs the reference {head:.3e}")
    assert err2 <= 1e-3 and head <= 1e-3


@pytest.mark.parametrize("name", ["base_4s", "large_4s", "tiny_group"])
def test_headline_arithmetic_against_reference_goldens(name):
    """The benchmark's own arithmetic END TO END -- bf16 encoder + fp16 vocoder with the fp16 activation stream and the
    fused ResBlock kernels -- against the outputs of the reference's fp32 modules (I_ea/predict.py:163-207): label
    agreement >= the stated floor, waveform RMS error <= 1e-3 (the north-star gate) on every sample the labels that
    agree can reach (the whole clip when all agree)."""
    c = load_case(name)
    out = _run(_engine(c, enc="bf16", voc="fp16"), c)
    agree, err = _check_wave(name + " bf16/fp16 (headline)", c, out, 1e-3)
    assert agree >= BF16_AGREE_FLOOR[name]
    assert bool(torch.isfinite(out["wave"]).all())


def test_nan_sample_does_not_fault_and_labels_stay_in_range():
    """A NaN sample in the clip (float WAVs can hold them) makes every feature NaN; torch.argmax then returns an in-range
    index (NaN counts as the maximum, first one wins -> 0).  The kernel must do the same instead of indexing the
    codebook with an uninitialised label."""
    c = load_case("tiny_group")
    m = c["meta"]
    eng = _engine(c)
    wave = c["wave"].clone()
    wave[1, 1234] = float("nan")
    pos = torch.tensor(c["frame_pos"], dtype=torch.int32, device="cuda")
    out = eng.predict_batch(wave.cuda(), c["mel"].cuda(), pos, m["lm"])
    torch.cuda.synchronize()
    lab = out["labels"].cpu()
    assert int(lab.min()) >= 0 and int(lab.max()) < m["K"]
    assert bool((lab[1] == 0).all())                                   # all-NaN similarities: first index, as torch.argmax
    assert np.array_equal(lab[0].numpy(), c["z"]["labels"][0]) and np.array_equal(lab[2].numpy(), c["z"]["labels"][2])


def test_mask_past_the_last_frame_yields_label_minus_one():
    """frame_pos + Lm beyond T: the reference fails on the slice-shape mismatch (I_ea/predict.py:166-168); the kernel
    marks such frames -1 and leaves the mel column alone, and predict_clips refuses the call on the host."""
    from speech_inpainting_amd.predict import check_mask_span
    c = load_case("tiny_group")
    m = c["meta"]
    eng = _engine(c)
    feats = eng.encode(c["wave"].cuda())
    pos = torch.tensor([m["T"] - 3] * m["B"], dtype=torch.int32, device="cuda")
    mel = c["mel"].cuda().clone()
    lab = eng.splice(feats, pos, 5, mel).cpu()
    assert bool((lab[:, :3] >= 0).all()) and bool((lab[:, 3:] == -1).all())
    assert torch.equal(mel[:, :, :m["T"] - 3].cpu(), c["mel"][:, :, :m["T"] - 3])
    with pytest.raises(ValueError, match="do not fit"):
        check_mask_span(eng, m["N"], 11264, [m["T"] - 3], 5)


def test_receiving_rank_weight_path_on_one_gpu():
    """The multi-GPU receive side without a second GPU: engine B allocates the packed blob
    (si_alloc_weights), the bytes arrive by a device copy standing in for the RCCL broadcast, and B must then compute
    bit-identically to the engine that read the checkpoint.  Also pins that the torch view aliases the library's blob
    and that the layout is a pure function of the model desc (equal size in both contexts)."""
    from speech_inpainting_amd.engine import InpaintingEngine
    c = load_case("tiny_group")
    m = c["meta"]
    for enc, voc in (("fp32", "fp32"), ("bf16", "fp16")):
        A = _engine(c, enc=enc, voc=voc)
        B = InpaintingEngine(c["harch"], c["varch"], m["K"], "cuda:0", enc, voc).alloc_weights()
        wa, wb = A.weights_tensor(), B.weights_tensor()
        pa, na = A.ctx.weights_ptr()
        pb, nb = B.ctx.weights_ptr()
        assert wb.data_ptr() == pb and wb.numel() == nb and wa.data_ptr() == pa and na == nb and pa != pb
        assert wb.dtype == torch.uint8 and wb.is_cuda
        wb.copy_(wa)
        torch.cuda.synchronize()
        oa, ob = _run(A, c), _run(B, c)
        assert torch.equal(oa["labels"], ob["labels"]) and torch.equal(oa["wave"], ob["wave"]) and torch.equal(oa["feats"], ob["feats"])
        del wa, wb
        A.ctx.close(); B.ctx.close()


def test_expected_inpaint_and_hifi_masked_batch_outputs():
    """predict_clips(diagnostics=True): the script's other two vocoder passes (I_ea/predict.py:123-128,177-189,198-201)
    as batch outputs, against the oracle's generator on the same mels."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.predict import predict_clips
    c = load_case("tiny_group")
    m = c["meta"]
    eng = _engine(c)
    n22 = m["N"] * 22050 // 16000
    w16 = [c["wave"][i].numpy() for i in range(m["B"])]
    w22 = [synth.synth_wave(1, n22, 70 + i, sr=22050)[0].numpy() for i in range(m["B"])]
    Tm = eng.ctx.mel_frames(n22)
    lm = 4
    pos = [2, 5, min(m["T"], Tm) - lm]
    tgt = torch.tensor([[1, 2, 3, 4], [5, 6, 7, 8], [99, 0, 50, 10]], dtype=torch.int64)
    out = predict_clips(eng, w16, w22, pos, lm, diagnostics=True, target_labels=tgt)
    torch.cuda.synchronize()
    mel = out["mel_masked"].cpu()
    hm = R.generator_forward(c["gsd"], c["varch"], R.extend_mel(mel))[:, 0, :]
    assert rms(out["hifi_masked"].cpu(), hm) <= 1e-5
    exp = mel.clone()
    for i in range(m["B"]):
        exp[i, :, pos[i]:pos[i] + lm] = c["cb"][tgt[i]].T
    ew = R.generator_forward(c["gsd"], c["varch"], R.extend_mel(exp))[:, 0, :]
    assert rms(out["expected_inpaint"].cpu(), ew) <= 1e-5
    assert out["cos_pred_target"].shape == (m["B"], lm) and bool(torch.isfinite(out["loss"]))


@pytest.mark.parametrize("name,voc,tol", [("tiny_group", "bf16x3", 1e-5), ("base_4s", "bf16x3", 1e-5), ("large_4s", "bf16x3", 1e-5),
                                          ("base_4s", "bf16", 1e-3), ("base_4s", "fp16", 2e-4), ("tiny_group", "fp16", 2e-4)])
def test_vocoder_split_bf16_modes(name, voc, tol):
    """bf16x3 (hi/lo split, 3 MFMAs per product) is the benchmark's vocoder arithmetic: it must stay fp32-equivalent
    (measured 1.5e-6 RMS at full size).  Plain bf16 is only required to meet the north-star gate (measured 7.4e-4)."""
    c = load_case(name)
    z = c["z"]
    out = _run(_engine(c, voc=voc), c)
    assert np.array_equal(out["labels"].numpy(), z["labels"])
    if "wave" in z.files:
        err = rms(out["wave"], z["wave"])
    else:
        err = max(rms(out["wave"][:, :2048], z["wave_head"]), rms(out["wave"][:, -2048:], z["wave_tail"]))
    print(f"{name} vocoder {voc}: waveform rms error {err:.3e}")
    assert err <= tol


def test_module_wrappers_keep_reference_signatures():
    from oracle import ref_cpu as R
    from speech_inpainting_amd.engine import CustomModel, Generator
    c = load_case("tiny_group")
    eng = _engine(c)
    x = R.mask_and_normalize(c["wave"], [0] * 3, [0] * 3)
    y = CustomModel(eng).eval()(x.cuda(), torch.ones_like(x, dtype=torch.int32).cuda()).cpu()
    ref = R.custom_model_forward(c["hsd"], c["harch"], x)
    assert y.shape == ref.shape and rms(y, ref) <= 2e-5 * rms(ref)
    mel = R.extend_mel(c["mel"])
    g = Generator(eng)
    g.remove_weight_norm()
    w = g(mel.cuda()).cpu()
    wref = R.generator_forward(c["gsd"], c["varch"], mel)
    assert w.shape == wref.shape and rms(w, wref) <= 1e-5


def test_batch_independence_and_determinism():
    """Clips are independent units: a clip's output must not depend on its batch neighbours, the vocoder chunking or
    the run (bit-exact), at the bench's own shape class (B > chunk)."""
    from speech_inpainting_amd import synth
    c = load_case("tiny_group")
    m = c["meta"]
    B = 7
    wave = synth.synth_wave(B, m["N"], 99).cuda()
    mel = synth.synth_mel(B, m["Tm"], 80, 98).cuda()
    pos = synth.synth_mask_frames(B, m["T"], m["lm"], 97).cuda()
    eng = _engine(c, chunk=3)
    a = eng.predict_batch(wave, mel, pos, m["lm"])
    b = eng.predict_batch(wave, mel, pos, m["lm"])
    assert torch.equal(a["wave"], b["wave"]) and torch.equal(a["labels"], b["labels"])
    one = eng.predict_batch(wave[4:5].contiguous(), mel[4:5].contiguous(), pos[4:5].contiguous(), m["lm"])
    assert torch.equal(one["labels"], a["labels"][4:5])
    assert torch.equal(one["wave"], a["wave"][4:5])


def test_errors_are_loud():
    from speech_inpainting_amd.native import NativeError
    c = load_case("tiny_group")
    eng = _engine(c)
    with pytest.raises((NativeError, ValueError)):
        eng.encode(torch.zeros(1, 100, device="cuda"))          # shorter than the receptive field
    from speech_inpainting_amd.engine import InpaintingEngine
    fresh = InpaintingEngine(c["harch"], c["varch"], 100, "cuda:0")
    with pytest.raises(NativeError):
        fresh.encode(torch.zeros(1, 8000, device="cuda"))       # forward before weights
    bad = dict(c["hsd"])
    bad.pop("final_layers.1.bias")
    with pytest.raises(NativeError, match="final_layers.1.bias"):
        InpaintingEngine(c["harch"], c["varch"], 100, "cuda:0").load_state(bad, c["gsd"], c["cb"])


def test_loss_half_matches_reference_goldens():
    """f-4: si_codebook_metrics through the C ABI vs outputs of the reference's own LossFunction (tests/golden/loss_metrics.npz)."""
    import os
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine, LossFunction
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_metrics.npz"))
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    for K in (100, 500):
        cb = synth.synth_codebook(K, 80, synth.DEFAULT_SEED + 2)
        eng = InpaintingEngine(harch, varch, K, "cuda:0", "fp32", "fp32").load_state(
            synth.synth_hubert_state(harch), synth.synth_generator_state(varch), cb)
        lf = LossFunction(eng)
        for tag in ("near", "cnear", "far"):
            values, labels = torch.from_numpy(g[f"{tag}_{K}_values"]), torch.from_numpy(g[f"{tag}_{K}_labels"])
            loss, pred = lf.cos_sim(values, labels)                          # same call as I_ea/predict.py:171
            cpt = lf.cos_sim_target_labels(pred, labels)                     # :172-173
            ref_loss = float(g[f"{tag}_{K}_loss"])
            assert np.array_equal(pred.cpu().numpy(), g[f"{tag}_{K}_pred"]), (K, tag)
            assert abs(float(loss) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (K, tag, float(loss), ref_loss)
            assert np.allclose(cpt.cpu().numpy(), g[f"{tag}_{K}_cos_pred_target"], atol=2e-6), (K, tag)
        # masked-frame form on (B, T, 80) features + an out-of-range target
        feats = torch.from_numpy(g[f"cnear_{K}_values"]).cuda()
        tgt = torch.from_numpy(g[f"cnear_{K}_labels"]).cuda()
        pos = torch.tensor([0, 2, 4, 6], dtype=torch.int32, device="cuda")
        m = eng.codebook_metrics(feats, pos, 3, tgt[:, :3].contiguous())
        assert m["pred_labels"].shape == (4, 3) and torch.isfinite(m["loss"])
        bad = tgt[:, :3].clone(); bad[1, 1] = K
        m2 = eng.codebook_metrics(feats, pos, 3, bad.contiguous())
        assert torch.isnan(m2["loss_terms"][1, 1]) and int(m2["pred_labels"][1, 1]) == -1


@pytest.mark.parametrize("tag", ["group", "layer"])
def test_padded_batches_match_reference_goldens(tag):
    """`CustomModel(input_values, attention_mask)` on a right-padded batch (I_ea/model.py:80-85) against the reference's own
    output on ALL frames (tests/golden/padded.npz): through the module wrapper with the processor's input_values, and
    through si_hubert_forward_padded with the raw clips (normalisation over the real samples fused into conv0)."""
    import os
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import CustomModel, InpaintingEngine
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "padded.npz"))
    lens = [int(n) for n in z["lens"]]
    seed = synth.DEFAULT_SEED
    harch = HubertArch.tiny() if tag == "group" else HubertArch.tiny(conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True)
    varch = VocoderArch.tiny()
    hsd = synth.synth_hubert_state(harch, seed)
    ref = torch.from_numpy(z[f"{tag}_feats"])
    waves = [synth.synth_wave(1, n, seed + 40 + i)[0] for i, n in enumerate(lens)]
    x, m = R.normalize_padded(waves)
    raw = torch.zeros(len(lens), max(lens))
    for b, w in enumerate(waves):
        raw[b, :len(w)] = w
    vl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    for enc, tol in (("fp32", 1e-4), ("bf16", 2e-2)):
        eng = InpaintingEngine(harch, varch, 100, "cuda:0", enc, "fp32").load_state(hsd, synth.synth_generator_state(varch), synth.synth_codebook(100))
        a = CustomModel(eng)(x.cuda(), m.cuda()).cpu()
        b = eng.encode(raw.cuda(), None, None, normalize=True, valid_len=vl).cpu()
        ea, eb = rms(a, ref) / rms(ref), rms(b, ref) / rms(ref)
        print(f"padded {tag} {enc}: wrapper {ea:.3e}, raw + valid_len {eb:.3e} (relative rms on all frames)")
        assert a.shape == ref.shape and ea <= tol and eb <= tol
        if enc == "fp32":
            # a full-length clip in a padded batch equals the same clip alone (its mask is all ones)
            alone = eng.encode(raw[0:1, :lens[0]].contiguous().cuda()).cpu()
            assert rms(alone[0], b[0]) <= 1e-5 * rms(ref)
            # a mask that is not right-padded is refused
            bad = m.clone(); bad[1, 10] = 0
            with pytest.raises(ValueError, match="right-padded"):
                CustomModel(eng)(x.cuda(), bad.cuda())


def test_mel_and_waveform_metrics_match_oracle():
    """f-4, the signal half: `Metrics.avg_cosine_sim / avg_d2_dist / rmse / sisdr` (I_ea/metrics.py:38-62,127-142) on the GPU
    against the oracle's restatement of those statements (parity unpinned: the reference module is not importable)."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.engine import Metrics
    c = load_case("tiny_group")
    eng = _engine(c)
    cb = c["cb"]
    met = Metrics(eng, cb.mean(dim=0))
    g = torch.Generator().manual_seed(3)
    for L in (1, 10, 57, 300):
        t1 = synth.synth_mel(1, L, 80, 200 + L)[0]
        t2 = t1 + 0.3 * torch.randn(80, L, generator=g)
        cos, d2, r = R.mel_signal_metrics(t1, t2, cb.mean(dim=0))
        assert abs(float(met.avg_cosine_sim(t1, t2)) - float(cos)) <= 2e-6
        assert abs(float(met.avg_d2_dist(t1, t2)) - float(d2)) <= 1e-5 * max(1.0, float(d2))
        assert abs(float(met.rmse(t1, t2)) - float(r)) <= 1e-5 * max(1.0, float(r))
    ref = synth.synth_wave(1, 22050, 5)[0].numpy()
    for noise in (0.5, 1e-2, 1e-4, 0.0):
        est = (0.7 * ref + noise * torch.randn(22050, generator=g).numpy()).astype(np.float32)
        want, got = R.sisdr(est, ref), met.sisdr(est, ref)
        print(f"sisdr noise {noise}: oracle {want:.4f} dB, GPU {got:.4f} dB")
        assert abs(got - want) <= (1e-3 if noise > 0 else 3.0)         # the noiseless case is eps / rounding residue in both
    batch = eng.ctx.mel_metrics(torch.stack([synth.synth_mel(1, 40, 80, 7)[0]] * 2).cuda(), torch.stack([synth.synth_mel(1, 40, 80, 8)[0]] * 2).cuda())
    assert batch.shape == (2, 3) and torch.equal(batch[0], batch[1])
