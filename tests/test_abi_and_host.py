"""CPU-side tests: the C-ABI library loads and exports every declared symbol; host logic (arch, checkpoint
flattening, sharding, config) behaves; no compute calls are made here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from speech_inpainting_amd import checkpoint, native, parallel, synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "si_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(si_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = native.load_library()
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/si_hip.h but not exported"
    assert set(names) == set(native.EXPORTS), (set(names) ^ set(native.EXPORTS))
    assert lib.si_version() == 3


def test_desc_struct_matches_header_size():
    # si_create rejects a mismatching struct_size; here only the Python mirror's arithmetic is checked
    n_i32 = 1 + 4 + 1 + 3 * 8 + 3 + 2 + 1 + 1 + 2 + 2 + 2 * 8 + 1 + 1 + 4 + 1 + 16 + 4
    assert ctypes.sizeof(native.ModelDesc) == 4 * n_i32
    d = native.make_desc(HubertArch.large(), VocoderArch.v1(), 500, "bf16", "bf16x3", 4)
    assert d.hidden_size == 1024 and d.feat_norm_layer == 1 and d.stable_layer_norm == 1 and d.conv_bias == 1
    assert d.rb_dilations[2][1] == 3 and d.up_kernels[3] == 4 and d.vocoder_math == 2 and d.num_clusters == 500


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.load_library(str(tmp_path / "libsi_hip.so"))


def test_cpu_device_is_rejected():
    from speech_inpainting_amd.engine import InpaintingEngine
    with pytest.raises(RuntimeError, match="GPU"):
        InpaintingEngine(HubertArch.tiny(), VocoderArch.tiny(), 100, "cpu")


def test_flatten_checkpoint_index_roundtrip():
    ha, va = HubertArch.tiny(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(ha), synth.synth_generator_state(va), synth.synth_codebook(100)
    blob, index = checkpoint.flatten_checkpoint(hsd, gsd, cb)
    lines = [l.split() for l in index.strip().split("\n")]
    names = [l[0] for l in lines]
    assert "codebook" in names and "generator.conv_pre.weight_g" in names and "final_layers.1.weight" in names
    assert len(names) == len(set(names))
    for l in lines:
        off, nd = int(l[1]), int(l[2])
        shape = [int(v) for v in l[3:3 + nd]]
        key = l[0]
        src = cb if key == "codebook" else (gsd[key[len("generator."):]] if key.startswith("generator.") else hsd[key])
        assert list(src.shape) == shape
        n = int(np.prod(shape)) if shape else 1
        assert np.array_equal(blob[off // 4: off // 4 + n], src.reshape(-1).numpy())


def test_hubert_key_normalisation_accepts_bare_hf_state():
    sd = {"feature_extractor.conv_layers.0.conv.weight": torch.zeros(2), "masked_spec_embed": torch.zeros(3),
          "hubert.encoder.layer_norm.weight": torch.zeros(1), "lm_head.weight": torch.zeros(1),
          "final_layers.0.weight": torch.zeros(1), "base_model.encoder.layer_norm.bias": torch.zeros(1)}
    out = checkpoint.normalize_hubert_keys(sd)
    assert set(out) == {"base_model.feature_extractor.conv_layers.0.conv.weight", "base_model.encoder.layer_norm.weight",
                        "final_layers.0.weight", "base_model.encoder.layer_norm.bias"}


def test_hub_name_is_refused_with_local_path_hint():
    with pytest.raises(FileNotFoundError, match="local path"):
        checkpoint.load_hubert_checkpoint("facebook/hubert-base-ls960")


def test_checkpoint_files_roundtrip(tmp_path):
    import json
    va = VocoderArch.v1()
    gsd = synth.synth_generator_state(VocoderArch.tiny())
    d = tmp_path / "LJ_V1"
    d.mkdir()
    torch.save({"generator": gsd}, d / "generator_v1")
    (d / "config.json").write_text(json.dumps(dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                                                   upsample_initial_channel=256, resblock_kernel_sizes=[3, 7, 11],
                                                   resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=80, sampling_rate=22050)))
    sd2, va2 = checkpoint.load_generator_checkpoint(str(d / "generator_v1"))
    assert va2.upsample_initial_channel == 256 and va2.resblock_dilation_sizes == va.resblock_dilation_sizes
    assert set(sd2) == set(gsd)
    np.save(tmp_path / "c.npy", synth.synth_codebook(100).numpy())
    assert checkpoint.load_codebook(str(tmp_path / "c.npy")).shape == (100, 80)
    import joblib
    from sklearn.cluster import MiniBatchKMeans
    km = MiniBatchKMeans(n_clusters=100)
    km.cluster_centers_ = synth.synth_codebook(100).numpy()
    joblib.dump(km, tmp_path / "model.km")
    assert torch.equal(checkpoint.load_codebook(str(tmp_path / "model.km")), synth.synth_codebook(100))
    hsd = synth.synth_hubert_state(HubertArch.tiny())
    torch.save(hsd, tmp_path / "save_checkpoint.pt")
    sd3, arch = checkpoint.load_hubert_checkpoint(str(tmp_path / "save_checkpoint.pt"), "base")
    assert set(sd3) == set(hsd) and arch.hidden_size == 768


def _write_hf_dir(path, arch, sd, fmt="safetensors"):
    import json
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(arch.to_hf(), f)
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    else:
        torch.save(dict(sd), os.path.join(path, "pytorch_model.bin"))


def test_local_huggingface_directory_loader(tmp_path):
    """The "HuggingFace checkpoint loader" of the north star: a LOCAL directory with config.json + model.safetensors (or
    pytorch_model.bin) stands in for `HubertModel.from_pretrained(name)` / `HubertConfig.from_pretrained(name)`
    (I_ea/model.py:26-40), with bare `HubertModel` keys and with the `hubert.`-prefixed keys of a `HubertForCTC` file
    (facebook/hubert-large-ls960-ft is one)."""
    assert HubertArch.from_hf_config(HubertArch.base().to_hf()) == HubertArch.base()
    assert HubertArch.from_hf_config(HubertArch.large().to_hf()) == HubertArch.large()
    ha = HubertArch.tiny()
    hsd = synth.synth_hubert_state(ha)
    bare = {k[len("base_model."):]: v for k, v in hsd.items() if k.startswith("base_model.")}
    bare["masked_spec_embed"] = torch.zeros(ha.hidden_size)                       # present in real files, unused in eval
    want = {k: v for k, v in hsd.items() if k.startswith("base_model.")}
    _write_hf_dir(str(tmp_path / "bare"), ha, bare)
    sd, arch = checkpoint.load_hubert_checkpoint(str(tmp_path / "bare"))
    assert arch == ha and set(sd) == set(want) and all(torch.equal(sd[k], want[k]) for k in want)
    ctc = {"hubert." + k: v for k, v in bare.items()}
    ctc["lm_head.weight"], ctc["lm_head.bias"] = torch.zeros(32, ha.hidden_size), torch.zeros(32)
    _write_hf_dir(str(tmp_path / "ctc"), ha, ctc, fmt="bin")
    sd2, arch2 = checkpoint.load_hubert_checkpoint(str(tmp_path / "ctc"))
    assert arch2 == ha and set(sd2) == set(want) and all(torch.equal(sd2[k], want[k]) for k in want)
    # the in-tree dump of the base config (I_ea/dataset/config.json:62-124) gives the base architecture
    with pytest.raises(FileNotFoundError, match="neither"):
        os.makedirs(tmp_path / "empty")
        checkpoint.load_hubert_checkpoint(str(tmp_path / "empty"))
    # a directory holds the encoder only; the head comes from the CustomModel .pt or is freshly initialised as the
    # reference's constructor initialises it (I_ea/model.py:75-78)
    head = checkpoint.fresh_final_layers(ha, seed=1234)
    assert head["final_layers.0.weight"].shape == (ha.hidden_size,) and head["final_layers.1.weight"].shape == (80, ha.hidden_size)
    assert float(head["final_layers.0.weight"].min()) == 1.0 and float(head["final_layers.1.weight"].abs().max()) <= ha.hidden_size ** -0.5 + 1e-6


def test_bench_spawns_its_own_ranks_when_world_size_is_unset(monkeypatch):
    """`python bench.py --gpus N` without a launcher: N ranks under torch.distributed.run as a CHILD process (never an exec),
    rendezvous on 127.0.0.1, before anything touches the GPU."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class _Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before the spawn")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    c = seen["cmd"]
    assert c[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in c and "127.0.0.1" in c
    assert c[-4:] == ["--gpus", "4", "--steps", "3"] and c[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_arch_from_hf_config_json():
    import json
    cfg = json.load(open("/root/reference/I_ea/dataset/config.json")) if os.path.exists("/root/reference/I_ea/dataset/config.json") else None
    if cfg is None:
        pytest.skip("reference tree not present on this machine")
    hub = cfg["hubert"] if "hubert" in cfg else cfg
    hub = next((v for v in cfg.values() if isinstance(v, dict) and v.get("model_type") == "hubert"), hub)
    a = HubertArch.from_hf_config(hub)
    assert a == HubertArch.base()


def test_bucket_by_length_groups_exact_lengths():
    from speech_inpainting_amd.predict import bucket_by_length
    lengths = [64000, 160000, 64000, 80000, 64000, 160000, 64000]
    b = bucket_by_length(lengths, max_batch=3)
    assert sorted(i for g in b for i in g) == list(range(len(lengths)))          # a partition
    assert all(len({lengths[i] for i in g}) == 1 and len(g) <= 3 for g in b)      # equal length, bounded size
    assert b == [[1, 5], [3], [0, 2, 4], [6]]                                     # longest first, input order inside
    assert bucket_by_length([], 4) == []
