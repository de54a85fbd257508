"""Worker of tests/test_gpu_multirank.py::test_rccl_accepts_the_library_owned_weight_blob: ONE rank, backend "nccl" (= RCCL on
ROCm).  Proves on the box what no gloo rehearsal can: that ProcessGroupNCCL initialises here (`device_id=` form), that
`dist.broadcast` accepts the `__cuda_array_interface__` view of the library-owned hipMalloc blob (si_weights_device_ptr), and
that the device-side metrics all-gather and the barrier run -- through speech_inpainting_amd/parallel.py's own functions, which
issue the collectives whenever a process group exists (world size 1 included).  Prints one line `NCCL_OK {...}`."""
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
    port = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    harch, varch = HubertArch.tiny(), VocoderArch.tiny()
    sd = (synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook())
    eng = parallel.setup_engine(lambda: InpaintingEngine(harch, varch, 100, dev, "bf16", "fp16"), lambda: sd, rank=0)   # load + broadcast(src = self)
    blob = eng.weights_tensor()
    before = blob.clone()
    ptr, nbytes = eng.ctx.weights_ptr()
    assert blob.data_ptr() == ptr and blob.numel() == nbytes
    # the receive side: a second engine's si_alloc_weights blob, filled by a device copy, then broadcast in place by RCCL
    recv = InpaintingEngine(harch, varch, 100, dev, "bf16", "fp16").alloc_weights()
    recv.weights_tensor().copy_(blob)
    parallel.broadcast_blob(recv.weights_tensor(), 0)
    torch.cuda.synchronize()
    same = bool(torch.equal(blob, before)) and bool(torch.equal(recv.weights_tensor(), before))
    G, N, lm = 3, 8000, 4
    Tm = mel_frames(N * 22050 // 16000)
    wave, mel = synth.synth_wave(G, N).to(dev), synth.synth_mel(G, Tm).to(dev)
    pos = synth.synth_mask_frames(G, harch.num_frames(N), lm).to(dev)
    a = eng.predict_batch(wave, mel, pos, lm)
    b = recv.predict_batch(wave, mel, pos, lm)
    m = parallel.gather_metrics([1.5, float(a["wave"].double().abs().sum())], dev)      # device-side all_gather
    parallel.barrier()
    torch.cuda.synchronize()
    ok = same and bool(torch.equal(a["wave"], b["wave"])) and m.is_cuda and tuple(m.shape) == (1, 2) and float(m[0, 0]) == 1.5
    print("NCCL_OK " + json.dumps({"ok": ok, "blob_bytes": nbytes, "backend": dist.get_backend(), "gathered": m.cpu().tolist()}), flush=True)
    dist.destroy_process_group()
    if not ok:
        raise SystemExit("RCCL path: blob or outputs changed")


if __name__ == "__main__":
    main()
