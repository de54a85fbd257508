"""Pin the CPU oracle (oracle/ref_cpu.py) to outputs of the reference's own modules
(tests/golden/*.npz, written by tools/make_goldens.py in the authoring container)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.common import GOLDEN, load_case, rms

torch.set_num_threads(8)


@pytest.mark.parametrize("name", ["tiny_group", "tiny_layer", "tiny_blind", "base_4s", "large_4s", "base_b4"])
def test_oracle_matches_reference(name):
    c = load_case(name)
    m, z = c["meta"], c["z"]
    out = R.predict_batch(c["hsd"], c["harch"], c["gsd"], c["varch"], c["cb"], c["wave"], c["mel"],
                          c["frame_pos"], m["lm"], blind=m["blind"])
    feats_ref = torch.from_numpy(z["feats"])
    assert out["feats"].shape == feats_ref.shape
    # fp32 restatement of the same ops: differences are summation-order only
    assert rms(out["feats"], feats_ref) <= 2e-5 * max(rms(feats_ref), 1.0)
    assert np.array_equal(out["labels"].numpy(), z["labels"])
    assert np.allclose(out["mel"].numpy(), z["mel_spliced"], rtol=0, atol=1e-6)
    if "wave" in z.files:
        w = torch.from_numpy(z["wave"])
        assert out["wave"].shape == w.shape
        assert rms(out["wave"], w) <= 1e-5
    else:
        assert rms(out["wave"][:, :2048], z["wave_head"]) <= 1e-5
        assert rms(out["wave"][:, -2048:], z["wave_tail"]) <= 1e-5
    if "wave_win" in z.files:        # the samples the spliced frames reach, per clip
        for i, lo in enumerate(z["wave_win_lo"]):
            assert rms(out["wave"][i, lo:lo + 16384], z["wave_win"][i]) <= 1e-5
    assert abs(rms(out["wave"]) - float(z["wave_rms"])) <= 1e-5


def test_base_b4_fixture_discriminates():
    """The base-size fixture whose labels are real decisions: >= 5 distinct codewords over its 4 x 10 masked frames, more than
    one inside a clip, and fp32 cosine margins that are neither degenerate nor huge."""
    import torch.nn.functional as F
    c = load_case("base_b4")
    z = c["z"]
    labels = z["labels"]
    assert labels.shape == (4, 10) and len(set(labels.reshape(-1).tolist())) >= 5
    assert sum(len(set(row.tolist())) > 1 for row in labels) >= 2
    feats = torch.from_numpy(z["feats"])
    _, cc = R.codebook_tables(c["cb"])
    v = torch.stack([feats[b, p:p + 10] for b, p in enumerate(c["frame_pos"])]).reshape(-1, 80)
    sim = F.cosine_similarity(v[:, None, :], cc[None], dim=-1)
    top2 = sim.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    assert float(margin.min()) > 0 and float(margin.median()) < 0.2


def test_normalize_matches_hf_processor():
    c = load_case("tiny_group")
    z, m = c["z"], c["meta"]
    sl = [R.mask_samples_from_frames(p, m["lm"]) for p in c["frame_pos"]]
    x = R.mask_and_normalize(c["wave"], [s for s, _ in sl], [l for _, l in sl])
    assert np.allclose(x[:, :64].numpy(), z["x_norm_head"], rtol=0, atol=2e-6)


def test_extend_mel_matches_interpolate(golden_dir):
    z = np.load(golden_dir + "/extend_mel.npz")
    for tm in (1, 2, 3, 7, 64, 200, 373, 500):
        x = torch.from_numpy(z[f"in_{tm}"])[None]
        y = R.extend_mel(x)[0].numpy()
        ref = z[f"out_{tm}"]
        assert y.shape == ref.shape, tm
        assert np.allclose(y, ref, rtol=0, atol=1e-6), tm


def test_shape_known_answers():
    """I/O contract pinned by I_ea/prediction/LJ050-0271/*.wav: 119 558 samples @16 kHz in ->
    164 352 = 642 x 256 samples @22.05 kHz out."""
    from speech_inpainting_amd.arch import mel_frames, extended_frames, HubertArch
    n22 = 164766
    tm = mel_frames(n22)
    assert tm == 373 and extended_frames(tm) == 642 and extended_frames(tm) * 256 == 164352
    assert HubertArch.base().feat_lengths(64000) == [64000, 12799, 6399, 3199, 1599, 799, 399, 199]
    assert mel_frames(88200) == 200 and extended_frames(200) == 344


def test_pcm_truncates_toward_zero():
    a = torch.tensor([0.99999, -0.99999, 0.5 / 32768 * 3, -0.5 / 32768 * 3])
    assert R.to_int16_pcm(a).tolist() == [32767, -32767, 1, -1]


def test_loss_half_matches_reference_lossfunction():
    """f-4: cos_sim loss / labels / cos_sim_target_labels of the oracle vs the reference's LossFunction outputs."""
    import os
    from speech_inpainting_amd import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_metrics.npz"))
    for K in (100, 500):
        cb = synth.synth_codebook(K, 80, synth.DEFAULT_SEED + 2)
        for tag in ("near", "cnear", "far"):
            values, labels = torch.from_numpy(g[f"{tag}_{K}_values"]), torch.from_numpy(g[f"{tag}_{K}_labels"])
            loss, pred, cpt = R.cos_sim_loss(values, labels, cb)
            assert np.array_equal(pred.numpy(), g[f"{tag}_{K}_pred"])
            assert abs(float(loss) - float(g[f"{tag}_{K}_loss"])) <= 1e-5 * max(1.0, abs(float(g[f"{tag}_{K}_loss"])))
            assert np.allclose(cpt.numpy(), g[f"{tag}_{K}_cos_pred_target"], atol=1e-6)


def test_padded_batches_match_reference_goldens():
    """Right-padded batches: the oracle's attention-mask branch (zeroed padded frames, excluded keys; modeling_hubert.py:
    428-437,664-689) and its restatement of the processor's padded normalisation, against outputs of the reference's
    `CustomModel.forward(input_values, attention_mask)` (tests/golden/padded.npz, tools/make_goldens.py::padded_cases)."""
    import os
    import numpy as np
    import torch
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "padded.npz"))
    lens = [int(n) for n in z["lens"]]
    seed = synth.DEFAULT_SEED
    for tag, harch in (("group", HubertArch.tiny()),
                       ("layer", HubertArch.tiny(conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True))):
        hsd = synth.synth_hubert_state(harch, seed)
        waves = [synth.synth_wave(1, n, seed + 40 + i)[0] for i, n in enumerate(lens)]
        assert np.allclose([float(hsd["final_layers.1.weight"][0, 0]), float(waves[1][100])], z[f"{tag}_probe"], atol=1e-7)
        x, m = R.normalize_padded(waves)
        assert np.allclose(x[:, :64].numpy(), z[f"{tag}_x_head"], atol=2e-6) and np.allclose(x[:, -64:].numpy(), z[f"{tag}_x_tail"], atol=2e-6)
        with torch.no_grad():
            feats = R.custom_model_forward(hsd, harch, x, attention_mask=m)
        ref = torch.from_numpy(z[f"{tag}_feats"])
        err = float((feats - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert feats.shape == ref.shape and err <= 2e-5, (tag, err)
        # the mask matters: without it the padded clips differ
        with torch.no_grad():
            nomask = R.custom_model_forward(hsd, harch, x)
        assert float((nomask[1] - ref[1]).pow(2).mean().sqrt()) > 1e-3


def test_f0_vqvae_restatement_matches_the_reference_modules():
    """Row f-2: `oracle.f0_encoder_forward` / `f0_vq_codes` against the outputs of the reference's OWN `Encoder`
    (I_da/src/modules/jukebox.py:200-262) and `Bottleneck` (vq.py:183-232) for seeded weights in the hubert_lut.json:36-52
    shape, T = 64 / 800 / 1000 F0 frames (tests/golden/f0_vqvae.npz, written by tools/make_goldens.py::f0_vqvae_cases)."""
    import numpy as np
    import torch
    from oracle import ref_cpu as R
    from speech_inpainting_amd import native, synth
    from tests.common import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, "f0_vqvae.npz"))
    sd = synth.synth_f0_vqvae_state(native.F0EncDesc(), 20, seed=11)
    assert np.allclose([float(sd["encoder.level_blocks.0.model.0.0.weight"][0, 0, 0]), float(sd["vq.level_blocks.0.k"][0, 0])], z["probe"], atol=1e-7)
    for T in (64, 800, 1000):
        h = R.f0_encoder_forward(sd, torch.from_numpy(z[f"f0_{T}"]))
        ref = torch.from_numpy(z[f"h_{T}"])
        assert h.shape == ref.shape == (2, 128, T // 16)
        assert float((h - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        codes = R.f0_vq_codes(h, sd["vq.level_blocks.0.k"])
        assert np.array_equal(codes.numpy(), z[f"codes_{T}"])                    # bit-exact indices
        assert len(set(z[f"codes_{T}"].reshape(-1).tolist())) >= 4               # the fixture exercises several bins


def test_resblock2_generator_matches_reference():
    """`Generator(h)` with `resblock: "2"` in the config_v3.json shape (I_ea/hifi_gan/models.py:52-73,89): the oracle against the
    reference's own module output (tests/golden/gen_v3.npz)."""
    import os
    import numpy as np
    import torch
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    from tests.common import GOLDEN
    z = np.load(os.path.join(GOLDEN, "gen_v3.npz"))
    varch = VocoderArch.v3()
    assert VocoderArch.from_config(dict(resblock="2", upsample_rates=[8, 8, 4], upsample_kernel_sizes=[16, 16, 8], upsample_initial_channel=256,
                                        resblock_kernel_sizes=[3, 5, 7], resblock_dilation_sizes=[[1, 2], [2, 6], [3, 12]])) == varch
    gsd = synth.synth_generator_state(varch, synth.DEFAULT_SEED + 1)
    mel = synth.synth_mel(2, 40, 80, synth.DEFAULT_SEED + 4)
    assert np.allclose([float(gsd["resblocks.0.convs.1.weight_v"][0, 0, 0]), float(mel[0, 0, 0])], z["probe"], atol=1e-7)
    assert not any("convs1" in k or "convs2" in k for k in gsd) and len(gsd) == 69
    with torch.no_grad():
        w = R.generator_forward(gsd, varch, mel)[:, 0, :]
    ref = torch.from_numpy(z["wave"])
    assert w.shape == ref.shape == (2, 40 * 256)
    assert float((w - ref).pow(2).mean().sqrt()) <= 1e-6


def test_hidden_state_restatement_matches_transformers_hidden_states():
    """Row f-2: the oracle's `hubert_get_feats` (fairseq `extract_features(output_layer=L)` restated; fairseq is absent) against
    `transformers.HubertModel(output_hidden_states=True).hidden_states[L]` on inputs prepared by the reference's own
    statements (tests/golden/hidden_layers.npz, tools/make_goldens.py::hidden_layer_cases)."""
    import json
    import os
    import numpy as np
    import torch
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch
    from tests.common import GOLDEN
    z = np.load(os.path.join(GOLDEN, "hidden_layers.npz"))
    meta = json.loads(str(z["meta"]))
    for tag, harch in (("group", HubertArch.tiny(num_hidden_layers=3)),
                       ("layer", HubertArch.tiny(num_hidden_layers=3, conv_bias=True, feat_extract_norm="layer", do_stable_layer_norm=True))):
        hsd = synth.synth_hubert_state(harch, meta["seed"] + 60)
        y = synth.synth_wave(2, meta["N"], meta["seed"] + 61).numpy().astype(np.float64)
        for b in range(2):
            for kind in ("clean", "masked"):
                sig = y[b] if kind == "clean" else R.ida_corrupt(y[b], meta["frame_start"], meta["mask_size"])
                for L in ((1, 2, 3) if tag == "group" else (1, 2)):
                    with torch.no_grad():
                        h = R.hubert_get_feats(hsd, harch, sig, L)
                    ref = torch.from_numpy(z[f"{tag}_{kind}_{b}_L{L}"])
                    assert h.shape == ref.shape
                    assert float((h - ref).pow(2).mean().sqrt()) <= 2e-6 * float(ref.pow(2).mean().sqrt())
    # the corruption: zero inside the span, y + 1e-6 (formed in float64) outside
    y0 = y[0]
    c = R.ida_corrupt(y0, 100, 50)
    assert c.dtype == np.float64 and (c[100:150] == 0).all() and np.array_equal(c[:100], y0[:100] + 1e-6) and np.array_equal(c[150:], y0[150:] + 1e-6)


def test_ida_length_bookkeeping_known_answers():
    """`match_length` + the `% (16 * 80)` tail removal of I_da/scripts/inpainting.py:219-255, product (engine.ida_match_lengths) and
    oracle, against a literal re-execution of the script's numpy statements."""
    import numpy as np
    from oracle import ref_cpu as R
    from speech_inpainting_amd.engine import ida_match_lengths

    def script(n_audio, n_code, n_f0):
        audio, code, code_inp, fo = np.zeros(n_audio), np.zeros(n_code), np.zeros(n_code), np.zeros((1, n_f0))
        hops = [1, 1, 320, 80]
        unit = np.lcm.reduce(hops)                                                  # multiseries.py:36
        fpu = [unit // h for h in hops]
        n_unit = min(s.shape[-1] // f for s, f in zip([audio, audio, code, fo], fpu))
        audio, code, fo = audio[: n_unit * fpu[0]], code[: n_unit * fpu[2]], fo[..., : n_unit * fpu[3]]
        to_remove = audio.shape[-1] % (16 * 80)                                     # inpainting.py:244
        assert to_remove % 320 == 0
        if to_remove:
            audio, code, code_inp, fo = audio[:-to_remove], code[: -(to_remove // 320)], code_inp[: -(to_remove // 320)], fo[..., : -(to_remove // 80)]
        return audio.shape[-1], code.shape[-1], code_inp.shape[-1], fo.shape[-1]

    assert ida_match_lengths(64000, 199, 797) == (62720, 196, 196, 784)
    for n_audio, n_code, n_f0 in ((64000, 199, 797), (64000, 199, 800), (160000, 499, 1997), (12800, 39, 157), (9600, 29, 117), (64000, 199, 700)):
        assert ida_match_lengths(n_audio, n_code, n_f0) == R.ida_match_lengths(n_audio, n_code, n_f0) == script(n_audio, n_code, n_f0)


def test_f0_encoder_restatement_shapes_and_known_answer():
    """The oracle's F0 VQ-VAE encoder (row f-2) against a direct evaluation of its definition on a
    hand-checkable case: all-zero convolution weights leave only the biases, so every res block adds its k1 bias and
    the last conv returns its bias; and T frames come out as T // 16."""
    import torch
    from oracle import ref_cpu as R
    from speech_inpainting_amd import native, synth
    desc = native.F0EncDesc()
    sd = synth.synth_f0_vqvae_state(desc, seed=2)
    assert R.f0_encoder_forward(sd, torch.randn(2, 1, 320)).shape == (2, 128, 20)
    z = {k: (torch.zeros_like(v) if k.endswith("weight") else v) for k, v in sd.items()}
    out = R.f0_encoder_forward(z, torch.randn(1, 1, 64))
    last = z["encoder.level_blocks.0.model.4.bias"]
    assert torch.allclose(out, last[None, :, None].expand_as(out))
    codes = R.f0_vq_codes(torch.randn(2, 128, 5), sd["vq.level_blocks.0.k"])
    assert codes.shape == (2, 5) and int(codes.min()) >= 0 and int(codes.max()) < 20


def test_slaney_filterbank_matches_transformers_librosa_compatible_one():
    """`oracle.mel_filterbank` restates `librosa.filters.mel(htk=False, norm='slaney')` (I_ea/dataset/mel_dump.py:66) from the
    published formula; librosa is not in the image.  `transformers.audio_utils.mel_filter_bank(norm='slaney',
    mel_scale='slaney')` is an independent implementation of the same filterbank (HuggingFace's feature extractors use it in
    librosa's place): the two must agree to rounding, for the path's parameters and for a second set (other sample rate,
    band count and edges)."""
    from transformers.audio_utils import mel_filter_bank
    for sr, n_fft, n_mels, fmin, fmax in [(22050, 1024, 80, 0.0, 8000.0), (16000, 400, 64, 50.0, 7600.0)]:
        theirs = mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax,
                                 sampling_rate=sr, norm="slaney", mel_scale="slaney")
        ours = np.asarray(R.mel_filterbank(sr, n_fft, n_mels, fmin, fmax), dtype=np.float64)
        assert theirs.shape == (1 + n_fft // 2, n_mels) and ours.shape == (n_mels, 1 + n_fft // 2)
        d = np.abs(theirs.T - ours).max()
        print(f"sr={sr} n_fft={n_fft} n_mels={n_mels}: max |ours - transformers| = {d:.3e} (filter peak {ours.max():.3e})")
        assert d <= 1e-7 * max(ours.max(), 1e-3) + 1e-9


def test_resampler_restatement_reproduces_the_reference_held_16k_file():
    """f-3: `librosa.load(path, sr=16000)` (I_ea/predict.py:79-80) = resampy's `kaiser_best` in the pinned librosa 0.9.1.  The
    reference holds ONE fixture for it: I_ea/hifi_gan/test_files/LJ001-0001_22k.wav and _16k.wav are the same utterance at both
    rates.  tests/golden/lj001_resample.npz (tools/make_resample_fixture.py) carries three excerpts of the two files' int16
    samples: the oracle's restatement, given the 22.05 kHz samples, must reproduce the 16 kHz samples BIT FOR BIT after the file's
    int16 quantisation (floor(32768 y)) -- the start of the file (left wing cut at sample 0), 1.8 s of its interior (with the
    whole-file time registers), and its end (right wing cut at the last sample, zeros behind)."""
    import json
    z = np.load(os.path.join(GOLDEN, "lj001_resample.npz"))
    meta = json.loads(str(z["meta"]))
    M = 80                                                        # 64 zero crossings / scale = 89 input = 65 output samples of missing neighbours
    total = 0
    for name, fo, fi, lo, hi in (("head", 0, 0, 0, M), ("seg", meta["start16"], meta["start22"], M, M), ("tail", meta["tail_start16"], meta["tail_start22"], M, 0)):
        x22, want = z[name + "22"], z[name + "16"]
        y = R.resample_kaiser_best(x22.astype(np.float32) / 32768.0, 22050, 16000, fix_length=False, first_output=fo, first_input=fi)
        n = min(len(y), len(want))
        q = np.floor(y[:n] * 32768.0)
        assert np.array_equal(q[lo:n - hi], want[lo:n - hi].astype(np.float64)), name
        total += n - hi - lo
        if name == "tail":                                        # resampy stops at int(n * ratio); the file holds zeros from there
            assert len(y) == meta["n_resampled"] - fo and not want[len(y):].any() and meta["zeros_after"]
    assert total > 34000
    # librosa's fix_length: one zero appended up to ceil(n * ratio)
    x = z["head22"].astype(np.float32) / 32768.0
    full = R.resample_kaiser_best(x, 22050, 16000)
    assert len(full) == int(np.ceil(len(x) * 16000 / 22050)) and len(full) - int(len(x) * 16000 / 22050) in (0, 1)
