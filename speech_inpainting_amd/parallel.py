"""One process per GPU; utterances are the unit of sharding.

Clips are independent (no cross-clip state; GroupNorm/LayerNorm are per-sample), so the batch is cut into
contiguous per-rank slices and the data path has NO collective.  The fabric is used exactly twice
(SURVEY.md section 8(e)): one broadcast of the packed weight blob from the rank that read the checkpoint, and
one all-gather of a small per-rank metrics vector.  With backend "nccl" these are RCCL collectives over xGMI;
the CPU tests run the same code over gloo.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend: str = "nccl") -> Tuple[int, int, int]:
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `n_items` utterances owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of the packed weight blob (uint8 view of si_weights_device_ptr, or any tensor).
    RCCL ("nccl") broadcasts device memory directly over xGMI.  The gloo backend exists only to rehearse N > 1 ranks
    without N GPUs (CPU tests; several ranks sharing one card): it has no device transport here, so a device blob is
    staged through host memory around the same collective."""
    if not dist.is_initialized():
        return blob
    if dist.get_backend() == "gloo" and dist.get_world_size() == 1:
        return blob                         # nothing to stage (the world-size-1 collective is kept for RCCL only, where it is what the box can test)
    if blob.is_cuda and dist.get_backend() == "gloo":
        host = blob.cpu()
        dist.broadcast(host, src=src)
        if dist.get_rank() != src:
            blob.copy_(host)
        return blob
    dist.broadcast(blob, src=src)
    return blob


def gather_metrics(values: Sequence[float], device="cpu") -> torch.Tensor:
    """All-gather a per-rank vector -> (world, len) float64 tensor on every rank."""
    v = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return v[None]
    out = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(out, v)
    return torch.stack(out)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def setup_engine(make_engine, load_checkpoint, rank: int, src: int = 0):
    """Rank `src` reads + packs the checkpoint; the others allocate the identically laid-out packed blob and
    receive it with one broadcast.  make_engine() -> engine; load_checkpoint() -> (hubert_sd, gen_sd, codebook)."""
    eng = make_engine()
    if rank == src:
        eng.load_state(*load_checkpoint())
    else:
        eng.alloc_weights()
    broadcast_blob(eng.weights_tensor(), src)
    eng.weights_check()                # the received bytes were packed for THIS context's layout (desc + SI_VOC_* options)
    return eng
