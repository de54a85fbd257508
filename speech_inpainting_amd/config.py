"""predict.yaml / config.yaml schema of the reference (I_ea/predict.yaml:1-60, I_ea/config.yaml:1-76).

Only the keys the predict script reads are interpreted (I_ea/predict.py:60-73,85-89,109,144-146,158-159); unknown keys
are ignored, so both shipped files load.  An optional `bench:` / `batch:` block may be added by users of this package.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Optional, Union

import torch
import yaml


class AttrDict(dict):
    """dict with attribute access (I_ea/hifi_gan/env.py:5-11)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def choose_device(idx: Union[int, str]) -> torch.device:
    """I_ea/utils.py:6-30: 'cpu' -> cpu; int -> that GPU, clamped to the last one; no GPU -> cpu.
    (The inpainting engine itself refuses a CPU device: there is no CPU path in this package.)"""
    if idx == "cpu":
        return torch.device("cpu")
    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        if n < int(idx) + 1:
            return torch.device(f"cuda:{n - 1}")
        return torch.device(f"cuda:{int(idx)}")
    return torch.device("cpu")


@dataclass
class PredictConfig:
    dataset: str
    wave_path: str
    save_pred: str
    n_clusters: int
    km_model_path: str            # .../km_model_<K>/model.km
    path2centroids: str
    device_index: Union[int, str]
    mask_start_sec: float
    mask_end_sec: float
    hifigan_checkpoint: str
    hifigan_config: str           # config.json beside the checkpoint (I_ea/predict.py:110-111)
    hubert_type: str
    hubert_checkpoint: str
    asr_model_name: Optional[str] = None
    raw: Optional[dict] = None

    # derived exactly as the script does (I_ea/predict.py:85-90)
    @property
    def mask_ms(self) -> int:
        return int((self.mask_end_sec - self.mask_start_sec) * 1000)

    @property
    def mask_frames(self) -> int:
        return self.mask_ms // 20

    @property
    def start_sample(self) -> int:
        return int(self.mask_start_sec * 16000)

    @property
    def end_sample(self) -> int:
        return int(self.mask_end_sec * 16000)

    @property
    def mask_pos(self) -> int:
        return self.start_sample // 320


def load_predict_config(path: str = "predict.yaml") -> PredictConfig:
    with open(path) as f:
        data = yaml.safe_load(f)
    try:
        ds = data["training_config"]["dataset"]
        n = int(data["km_model"]["n_clusters"])
        ck = data["hifi_gan"]["checkpoint_file"]
        mask = data.get("mask", {})
        return PredictConfig(
            dataset=ds,
            wave_path=data["wave"][ds]["wave_path"],
            save_pred=data["wave"][ds]["save_pred"],
            n_clusters=n,
            km_model_path=os.path.join(data["km_model"][ds]["km_model_path"], f"km_model_{n}/model.km"),
            path2centroids=os.path.join(data["km_model"][ds]["path2centroids"], f"km_model_{n}/label_dir/validation"),
            device_index=data.get("device", {}).get("index", 0),
            mask_start_sec=float(mask.get("start_pos_in_sec", 0.0)),
            mask_end_sec=float(mask.get("end_pos_in_sec", 0.0)),
            hifigan_checkpoint=ck,
            hifigan_config=os.path.join(os.path.split(ck)[0], "config.json"),
            hubert_type=str(data["hubert_model"]["type"]),
            hubert_checkpoint=data["hubert_model"][ds]["model_checkpoint"],
            asr_model_name=data.get("ASR_model", {}).get("model_name"),
            raw=data)
    except KeyError as e:
        raise KeyError(f"{path}: missing key {e} (schema: I_ea/predict.yaml)") from None


def load_hifigan_config(path: str) -> AttrDict:
    with open(path) as f:
        return AttrDict(json.load(f))
