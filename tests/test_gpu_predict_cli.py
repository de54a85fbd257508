"""End-to-end: the predict.py entry point on synthetic checkpoint FILES in the reference's formats
(CustomModel .pt, HiFi-GAN generator dict + config.json, joblib k-means), reading the reference's YAML schema."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_predict_entry_point_writes_reference_outputs(tmp_path, monkeypatch):
    import joblib
    from scipy.io import wavfile
    from sklearn.cluster import MiniBatchKMeans
    from oracle import ref_cpu as R
    from speech_inpainting_amd import audio, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.predict import main

    harch, varch = HubertArch.base(), VocoderArch.v1()
    hsd, gsd, cb = synth.synth_hubert_state(harch, pos_conv_style="legacy"), synth.synth_generator_state(varch), synth.synth_codebook(100)
    (tmp_path / "trained_models").mkdir()
    torch.save(dict(hsd), tmp_path / "trained_models" / "save_checkpoint.pt")
    (tmp_path / "hifi_gan" / "LJ_V1").mkdir(parents=True)
    torch.save({"generator": dict(gsd)}, tmp_path / "hifi_gan" / "LJ_V1" / "generator_v1")
    (tmp_path / "hifi_gan" / "LJ_V1" / "config.json").write_text(json.dumps(dict(
        resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
        resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=80, sampling_rate=22050, seed=1234)))
    kdir = tmp_path / "kmeans" / "km_model_100"
    kdir.mkdir(parents=True)
    km = MiniBatchKMeans(n_clusters=100)
    km.cluster_centers_ = cb.numpy()
    joblib.dump(km, kdir / "model.km")
    w22 = synth.synth_wave(1, 66150, 5, sr=22050)[0].numpy()          # 3 s at 22.05 kHz
    (tmp_path / "wavs").mkdir()
    wavfile.write(tmp_path / "wavs" / "clip.wav", 22050, (w22 * 32767).astype(np.int16))
    (tmp_path / "predict.yaml").write_text(f"""
training_config: {{dataset: LJSpeech}}
wave: {{LJSpeech: {{wave_path: '{tmp_path}/wavs/clip.wav', save_pred: '{tmp_path}/prediction'}}}}
mask: {{start_pos_in_sec: 1.2, end_pos_in_sec: 1.5}}
device: {{index: 0}}
hifi_gan: {{checkpoint_file: '{tmp_path}/hifi_gan/LJ_V1/generator_v1'}}
hubert_model: {{type: 'base', LJSpeech: {{model_checkpoint: '{tmp_path}/trained_models/save_checkpoint.pt'}}}}
km_model: {{n_clusters: 100, LJSpeech: {{path2centroids: '{tmp_path}/kmeans/', km_model_path: '{tmp_path}/kmeans/'}}}}
""")
    monkeypatch.chdir(tmp_path)
    assert main([]) == 0
    out = tmp_path / "prediction" / "clip"
    for f in ("orig.wav", "masked.wav", "hifi_masked.wav", "inpainted.wav"):
        assert (out / f).exists(), f
    sr, pcm = wavfile.read(out / "inpainted.wav")
    assert sr == 22050 and pcm.dtype == np.int16

    # same glue, oracle in place of the engine (mel front-end included): int16 samples differ by a couple of LSB at
    # most (the HIP mel is within ~1e-5 of torch's FFT-based one; 1 LSB = 3e-5)
    # `librosa.load(path, sr=16000)` / `sr=22050` (I_ea/predict.py:79-80): the oracle's restatement of librosa 0.9.1's resampler
    raw, sr_file = audio.read_wav(str(tmp_path / "wavs" / "clip.wav"))
    assert sr_file == 22050
    w16 = R.resample_kaiser_best(raw, 22050, 16000).astype(np.float32)
    w22r = raw
    from speech_inpainting_amd.config import load_predict_config
    cfg = load_predict_config(str(tmp_path / "predict.yaml"))
    pos, lm = cfg.mask_pos, cfg.mask_frames
    mel = R.masked_mel([w22r], [cfg.start_sample * 22050 // 16000], [cfg.end_sample * 22050 // 16000])   # predict.py:99-106
    ref = R.predict_batch(hsd, harch, gsd, varch, cb, torch.from_numpy(w16)[None], mel, [pos], lm)
    ref_pcm = audio.to_int16_pcm(ref["wave"][0])
    assert pcm.shape == ref_pcm.shape == (mel.shape[2] * 441 // 256 * 256,) or pcm.shape == ref_pcm.shape
    diff = np.abs(pcm.astype(np.int32) - ref_pcm.astype(np.int32))
    assert diff.max() <= 2 and (diff > 1).mean() < 0.001, (diff.max(), (diff > 0).mean())
    sr16, masked = wavfile.read(out / "masked.wav")
    assert sr16 == 16000 and np.all(masked[pos * 320 + 80:(pos + lm) * 320 - 1] == 0)
