"""CPU tests of the host glue around the path: YAML schema, mel front-end shape/known answers, PCM conversion."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from speech_inpainting_amd import audio, config
from speech_inpainting_amd.arch import mel_frames

REF_YAML = "/root/reference/I_ea/predict.yaml"


def _yaml_text():
    return """
training_config: {dataset: LJSpeech}
wave: {LJSpeech: {wave_path: './w/LJ035-0125.wav', save_pred: './prediction/LJSpeech'}}
mask: {start_pos_in_sec: 1.8, end_pos_in_sec: 2.0}
device: {index: 0}
hifi_gan: {checkpoint_file: './hifi_gan/LJ_V1/generator_v1'}
hubert_model: {type: 'base', LJSpeech: {model_checkpoint: './trained_models/save_checkpoint.pt'}}
ASR_model: {cache_dir: './pretrained_models', model_name: 'openai/whisper-small'}
km_model: {n_clusters: 100, LJSpeech: {path2centroids: './dataset/kmeans/LJSpeech/', km_model_path: './dataset/kmeans/LJSpeech/'}}
unknown_block: {ignored: true}
"""


def test_predict_yaml_schema(tmp_path):
    p = tmp_path / "predict.yaml"
    p.write_text(_yaml_text())
    c = config.load_predict_config(str(p))
    assert c.mask_ms == 199 or c.mask_ms == 200          # int((2.0-1.8)*1000) is 199 in floating point, as in the script
    assert c.mask_frames == c.mask_ms // 20 and c.mask_pos == int(1.8 * 16000) // 320 == 90
    assert c.km_model_path.endswith("km_model_100/model.km") and c.hifigan_config.endswith("LJ_V1/config.json")
    assert c.hubert_type == "base" and c.device_index == 0


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference tree not present")
def test_reference_predict_yaml_loads():
    c = config.load_predict_config(REF_YAML)
    assert c.hubert_type == "large" and c.n_clusters == 100 and c.mask_frames == 20


def test_missing_key_is_named(tmp_path):
    p = tmp_path / "predict.yaml"
    p.write_text(_yaml_text().replace("hifi_gan:", "hifi_gann:"))
    with pytest.raises(KeyError, match="hifi_gan"):
        config.load_predict_config(str(p))


def test_choose_device_semantics():
    assert config.choose_device("cpu").type == "cpu"
    d = config.choose_device(7)                     # clamps to the last GPU, or falls back to cpu (I_ea/utils.py:22-30)
    assert d.type in ("cpu", "cuda")


def test_oracle_mel_frontend_shape_and_filterbank():
    """Known-answer properties of the restated librosa filterbank (librosa is absent: 'filterbank parity unpinned')."""
    fb = R.mel_filterbank()
    assert fb.shape == (80, 513) and fb.dtype == np.float32 and (fb >= 0).all()
    # Slaney area normalisation: each triangle integrates to ~1 on the Hz axis (bin spacing 22050/1024)
    area = fb.sum(axis=1) * (22050 / 1024)
    assert np.allclose(area[5:], 1.0, atol=0.12)
    assert fb[:, 372:].max() == 0.0                  # nothing above fmax = 8000 Hz (bin 371.5)
    for n in (88200, 164766, 22050):
        y = torch.randn(2, n).clamp(-1, 1) * 0.3
        m = R.mel_spectrogram(y)
        assert m.shape == (2, 80, mel_frames(n))
        assert float(m.min()) >= np.log(1e-5) - 1e-6
    z = R.mel_spectrogram(torch.zeros(1, 4410))
    assert torch.allclose(z, torch.full_like(z, float(np.log(1e-5))))


def test_pcm_truncation_and_peak_normalise():
    a = torch.tensor([0.99999, -0.99999, 1.5 / 32768, -1.5 / 32768, 1.0, -1.0, 0.5, -0.25])
    with np.errstate(invalid="ignore"):
        ref = (a * 32768).cpu().numpy().astype("int16")           # the reference's two statements, I_ea/predict.py:204-206
    got = audio.to_int16_pcm(a)
    defined = np.array([0, 1, 2, 3, 5, 6, 7])                      # 32768.0 -> int16 (index 4) is undefined behaviour in the reference's cast
    assert np.array_equal(got[defined], ref[defined])              # bit-exact wherever the reference's result is defined
    assert got[4] == 32767                                         # ... and pinned to full-scale positive where it is not
    assert ref[:4].tolist() == [32767, -32767, 1, -1] and ref[5:].tolist() == [-32768, 16384, -8192]
    assert audio.to_int16_pcm(a, clip=True).tolist() == [32767, -32767, 1, -1, 32767, -32768, 16384, -8192]
    with np.errstate(invalid="ignore"):
        assert np.array_equal(got[defined], R.to_int16_pcm(a)[defined])
    x = np.array([0.1, -0.5, 0.25], dtype=np.float32)
    assert np.allclose(R.peak_normalize_095(x), x / 0.5 * 0.95)
    assert np.array_equal(R.peak_normalize_095(np.zeros(4, np.float32)), np.zeros(4, np.float32))


def test_resample_lengths_and_tone():
    sr = 22050
    t = np.arange(sr) / sr
    x = np.sin(2 * np.pi * 440 * t).astype(np.float32)
    y = audio.resample(x, sr, 16000)
    assert abs(len(y) - 16000) <= 1
    ref = np.sin(2 * np.pi * 440 * np.arange(len(y)) / 16000)
    assert np.sqrt(np.mean((y[200:-200] - ref[200:-200]) ** 2)) < 1e-3
