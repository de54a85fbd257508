"""N > 1 ranks with the REAL engine, rehearsed on the one GPU of the test box: rank 0 reads the
checkpoint, the other ranks allocate the packed blob (si_alloc_weights) and receive it by the broadcast, every rank runs
its utterance shard, metrics come back by the all-gather.  The 8-GPU run differs only in the backend string."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(nproc, script_args, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: torchrun binds its own rendezvous port (no bind / close / reuse race between back-to-back launches)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={nproc}"] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("nproc,enc,voc", [(2, "fp32", "fp32"), (3, "bf16", "fp16")])
def test_ranks_share_one_gpu_and_match_the_single_batch_run(nproc, enc, voc):
    p = _launch(nproc, [os.path.join(ROOT, "tests", "rank_worker.py"), enc, voc])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("RANKS_OK ")][-1]
    r = json.loads(line[len("RANKS_OK "):])
    assert r["ok"] and r["world"] == nproc and sum(r["clips"]) == 7


def test_rccl_accepts_the_library_owned_weight_blob():
    """backend="nccl", world size 1: ProcessGroupNCCL init with device_id, dist.broadcast on the __cuda_array_interface__ view of the
    library's hipMalloc blob, device-side all_gather and barrier -- issued by parallel.py's own functions (tests/nccl_worker.py)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py"), str(_free_port())], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("NCCL_OK ")][-1][len("NCCL_OK "):])
    assert r["ok"] and r["backend"] == "nccl" and r["blob_bytes"] > 0


def test_bench_gpus_flag_without_a_launcher_spawns_the_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver starts the N = 1 line) must launch its own ranks as
    child processes and relay rank 0's JSON line (gloo here: two ranks share this box's one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SI_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-fp32-leg", "--cpu-clips", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 8 and r["value"] > 0


def test_bench_two_rank_rehearsal_prints_the_contract_line():
    """bench.py --gpus 2 launched exactly as the driver launches it, but with SI_DIST_BACKEND=gloo so two ranks can share
    this box's GPU (small batch: the point is the rendezvous, the broadcast into si_alloc_weights memory, the barrier /
    max-over-ranks timing and the JSON line)."""
    env_backend = os.environ.get("SI_DIST_BACKEND")
    os.environ["SI_DIST_BACKEND"] = "gloo"
    try:
        p = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-fp32-leg", "--cpu-clips", "0"])
    finally:
        if env_backend is None:
            os.environ.pop("SI_DIST_BACKEND", None)
        else:
            os.environ["SI_DIST_BACKEND"] = env_backend
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 8 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["steps"] == 2 and "roofline" in r
    # BASELINE's metric names the PER-GPU figure: `value` is the whole-job aggregate the driver's contract asks for,
    # `value_per_gpu` = the clips of the slowest rank x 4 s / wall -- with equal shards exactly value / n_gpus
    assert "per GPU" in r["metric"] and r["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert r["value_per_gpu"] > 0 and abs(r["value_per_gpu"] * 2 - r["value"]) <= 0.02 * r["value"]
    assert "whole-job" in r["unit"] and "configs4" not in r            # the single-GPU config legs do not run at N > 1
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rehearsal_2rank.json"), "w") as f:
        f.write(line + "\n")
