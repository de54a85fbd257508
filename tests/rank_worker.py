"""Worker of tests/test_gpu_multirank.py: one rank of a torch.distributed.run launch in which SEVERAL ranks share the one
GPU of the box (gloo backend: RCCL refuses two ranks on one device).  Every rank builds the REAL engine through
parallel.setup_engine -- rank 0 reads and packs the checkpoint, the others si_alloc_weights and receive the packed blob
by the broadcast -- runs its utterance shard, and rank 0 compares the gathered per-clip checksums with the same clips
run in one batch on its own engine.  Prints one line `RANKS_OK {...}` on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from speech_inpainting_amd import parallel, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames
    from speech_inpainting_amd.engine import InpaintingEngine
    enc, voc = sys.argv[1], sys.argv[2]
    rank, local_rank, world = parallel.init_distributed("gloo")        # before any GPU call
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    loads = []

    def checkpoint():
        loads.append(1)
        return synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook()

    eng = parallel.setup_engine(lambda: InpaintingEngine(harch, varch, 100, dev, enc, voc), checkpoint, rank)
    assert len(loads) == (1 if rank == 0 else 0)
    G, N, lm = 7, 8000, 4
    Tm = mel_frames(N * 22050 // 16000)
    wave, mel = synth.synth_wave(G, N).to(dev), synth.synth_mel(G, Tm).to(dev)
    pos = synth.synth_mask_frames(G, harch.num_frames(N), lm).to(dev)
    lo, hi = parallel.shard_range(G, rank, world)
    out = eng.predict_batch(wave[lo:hi].contiguous(), mel[lo:hi].contiguous(), pos[lo:hi].contiguous(), lm)
    torch.cuda.synchronize()
    # per-clip checksums padded to the largest shard: (clips, sum |wave| in fp64, sum of labels)
    per = -(-G // world)
    vec = [float(hi - lo)]
    for i in range(per):
        ok = i < hi - lo
        vec += [float(out["wave"][i].double().abs().sum()) if ok else 0.0, float(out["labels"][i].sum()) if ok else 0.0]
    m = parallel.gather_metrics(vec, "cpu")
    parallel.barrier()
    if rank == 0:
        full = eng.predict_batch(wave, mel, pos, lm)
        torch.cuda.synchronize()
        got_w, got_l = [], []
        for r in range(world):
            n = int(m[r, 0])
            for i in range(n):
                got_w.append(float(m[r, 1 + 2 * i])); got_l.append(float(m[r, 2 + 2 * i]))
        ref_w = [float(full["wave"][i].double().abs().sum()) for i in range(G)]
        ref_l = [float(full["labels"][i].sum()) for i in range(G)]
        ok = got_w == ref_w and got_l == ref_l
        print("RANKS_OK " + json.dumps({"ok": ok, "world": world, "clips": [int(m[r, 0]) for r in range(world)],
                                        "enc": enc, "voc": voc}), flush=True)
        if not ok:
            raise SystemExit(f"sharded outputs differ from the one-batch run: {got_w} vs {ref_w}; {got_l} vs {ref_l}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
