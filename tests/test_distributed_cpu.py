"""World-size-2 gloo rehearsal of the multi-GPU path: utterance sharding, weight-blob broadcast, metrics all-gather.
The compute step is the CPU oracle standing in for the engine (tests may use the oracle as the checker); what is
under test is the host logic in speech_inpainting_amd/parallel.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEngine:
    """Stands in for InpaintingEngine: a 'packed blob' that rank 0 fills and the others receive."""

    def __init__(self):
        self.blob = None

    def load_state(self, hsd, gsd, cb):
        self.blob = torch.arange(1000, dtype=torch.uint8) * 3
        return self

    def alloc_weights(self):
        self.blob = torch.zeros(1000, dtype=torch.uint8)
        return self

    def weights_tensor(self):
        return self.blob

    def weights_check(self):
        self.checked = True


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from speech_inpainting_amd import parallel, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames
    from oracle import ref_cpu as R
    torch.set_num_threads(2)
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    loads = []
    eng = parallel.setup_engine(_FakeEngine, lambda: (loads.append(1), None, None), rank)
    assert len(loads) == (1 if rank == 0 else 0)              # only the source rank reads the checkpoint
    assert torch.equal(eng.weights_tensor(), torch.arange(1000, dtype=torch.uint8) * 3)
    assert eng.checked                                            # the layout check runs on every rank after the broadcast
    # shard a global list of 5 utterances; each rank runs its slice; outputs must equal the 1-process run
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook()
    G, N, lm = 5, 6400, 4
    Tm = mel_frames(N * 22050 // 16000)
    wave, mel = synth.synth_wave(G, N), synth.synth_mel(G, Tm)
    pos = synth.synth_mask_frames(G, harch.num_frames(N), lm).tolist()
    lo, hi = parallel.shard_range(G, rank, world)
    out = R.predict_batch(hsd, harch, gsd, varch, cb, wave[lo:hi], mel[lo:hi], pos[lo:hi], lm)
    m = parallel.gather_metrics([float(hi - lo), float(out["wave"].double().pow(2).sum()), float(out["labels"].sum())])
    parallel.barrier()
    if rank == 0:
        full = R.predict_batch(hsd, harch, gsd, varch, cb, wave, mel, pos, lm)
        q.put((m.tolist(), float(full["wave"].double().pow(2).sum()), float(full["labels"].sum())))
    dist.destroy_process_group()


def test_two_rank_sharding_broadcast_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    m, energy, labsum = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [row[0] for row in m] == [3.0, 2.0]                                  # 5 utterances over 2 ranks
    assert abs(sum(row[1] for row in m) - energy) <= 1e-6 * energy              # per-utterance outputs are unchanged by sharding
    assert sum(row[2] for row in m) == labsum
