#!/usr/bin/env python3
"""Benchmark of the I_ea predict hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident batch of synthetic clips: raw 16 kHz + 22.05 kHz waveforms ->
masked log-mel front-end; masked 16 kHz waveform -> HuBERT-base encoder -> LN+Linear head -> codeword arg-max + mel
splice -> x441/256 stretch -> HiFi-GAN V1 -> waveforms.  Workload at every N: BASELINE.json configs[1] per GPU (batch 32 x 4 s clips, 200 ms mask, HuBERT-base
encoder GEMMs on bf16 MFMA with fp32 accumulate); at N > 1 utterances are sharded, 32 per rank (configs[2] at N = 8),
weights arrive by one RCCL broadcast, metrics by one all-gather.  Inputs and weights are in HBM before the timed
region; outputs stay in HBM.

Arithmetic of the headline number: encoder GEMMs and attention products on bf16 MFMA (what configs[1] names; softmax,
LayerNorm, head and arg-max in fp32); vocoder convolutions on fp16 MFMA with fp32 accumulate and fp16 activation storage ("validated mixed":
operands and stored activations rounded to fp16 -- an 11-bit significand, 8x finer than bf16 -- saturating at +-65504;
bias / residual / MRF arithmetic in fp32 on the accumulators; measured waveform RMS error vs the reference 1.35e-4
against the north-star gate of 1e-3, tests/test_gpu_parity.py).  Two shorter legs on the same inputs are reported in
the same JSON line: "bf16x3_vocoder" (every fp32 operand split into bf16 hi + lo, three MFMAs per product: 1.5e-6
RMS, fp32-equivalent) and "fp32_vocoder" (exact fp32 MFMA), and `vocoder_check` holds the live RMS difference between
the headline waveform and the fp32 leg's.

Prints ONE JSON line on rank 0 (metric/value/... plus `roofline` for the dominant kernel family, measured with HIP
events inside the timed region, and `cpu_baseline`: the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

CLIP_SECONDS = 4.0
N_SAMPLES = 64000
MASK_FRAMES = 10          # 200 ms
GFLOP_PER_CLIP = 268.3    # algorithmic, BASELINE.md section 2
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0, "f16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks
MFMA_PER_PRODUCT = {"f32": 1, "bf16": 1, "bf16x3": 3, "f16": 1}
# What a dense 16-bit MFMA stream SUSTAINS on this chip (the clock falls under matrix load): measured with
# tools/ubench/mfma_rate.hip on every CU, random operands -- v_mfma_f32_16x16x32 1.82 PFLOP/s (registers only or LDS-fed),
# v_mfma_f32_32x32x16 1.48 PFLOP/s from registers, 1.37 LDS-fed (profiles/r02_ubench_mfma_rate.txt).  Reported beside the
# nominal peak, never instead of it.
SUSTAINED_MFMA_TFLOPS = {"bf16": 1820.0, "f16": 1820.0, "bf16x3": 1820.0}
PEAK_HBM_GBS = 8000.0                                             # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def family_math(name: str) -> str:
    for m in ("bf16x3", "bf16", "f16", "f32"):
        if f"_{m}" in name:
            return m
    return "f32"


def host_cpu_share():
    """(logical CPUs of the host, CPUs this process may run on): the smaller of its affinity mask and its cgroup CPU
    quota -- a 1-GPU box of this pool gives a job 16 of the host's CPUs, and torch threads beyond that share only
    contend with each other."""
    total = os.cpu_count() or 1
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else total
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    share = min(share, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    share = min(share, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
        except Exception:
            pass
    return total, max(share, 1)


def cpu_baseline(batch: int, budget_s: float = 45.0):
    """the CPU oracle (plain torch fp32 restatement of the reference, resident tensors, mel
    front-end included) on every CPU this process may use, 1 warm-up + min of 5 passes at batch 1 and at batch `batch`
    (8).  Bounded: the pass count at the larger batch shrinks (never below 2) so the whole leg stays under budget_s."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames
    total, cores = host_cpu_share()
    torch.set_num_threads(cores)
    harch, varch = HubertArch.base(), VocoderArch.v1()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook()
    Tm = mel_frames(N_SAMPLES * 22050 // 16000)
    wave = synth.synth_wave(batch, N_SAMPLES)
    wave22 = synth.synth_wave(batch, N_SAMPLES * 22050 // 16000, synth.DEFAULT_SEED + 6, sr=22050).numpy()
    pos = synth.synth_mask_frames(batch, harch.num_frames(N_SAMPLES), MASK_FRAMES).tolist()
    s22 = [p * 320 * 22050 // 16000 for p in pos]
    e22 = [(p + MASK_FRAMES) * 320 * 22050 // 16000 for p in pos]
    assert R.masked_mel(wave22[:1], s22[:1], e22[:1]).shape[2] == Tm

    def one_pass(n):
        t0 = time.perf_counter()
        R.predict_batch(hsd, harch, gsd, varch, cb, wave[:n], R.masked_mel(wave22[:n], s22[:n], e22[:n]), pos[:n], MASK_FRAMES)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    one_pass(1)                                              # warm-up
    t1 = [one_pass(1) for _ in range(5)]
    tb, reps = [], 0
    if batch > 1:
        first = one_pass(batch)                              # warm-up at the larger batch
        left = budget_s - (time.perf_counter() - t_begin)
        reps = max(2, min(5, int(left / max(first, 1e-3))))
        tb = [one_pass(batch) for _ in range(reps)]
    best1, bestb = min(t1), (min(tb) if tb else None)
    v1 = CLIP_SECONDS / best1
    vb = batch * CLIP_SECONDS / bestb if bestb else v1
    return {"value": round(max(v1, vb), 3), "unit": "x real-time (audio-sec/wall-sec)", "cores": cores, "kind": "port",
            "host_logical_cpus": total, "threads_used": cores,
            "batch1": {"value": round(v1, 3), "best_s": round(best1, 3), "passes": 5},
            f"batch{batch}": {"value": round(vb, 3), "best_s": round(bestb, 3) if bestb else None, "passes": reps},
            "sample": f"torch CPU oracle (oracle/ref_cpu.py), fp32, resident tensors, 1 warm-up + min of 5 passes at batch 1 "
                      f"and min of {reps} at batch {batch} of the same 4 s clips; {cores} torch threads = this job's CPU share "
                      f"of the host's {total} logical CPUs; {time.perf_counter() - t_begin:.1f} s of CPU work in all"}


def measured_traffic(family: str):
    """HBM bytes per launch of a kernel family from the newest committed PMC table (profiles/rNN_hbm_traffic.json,
    written by tools/collect_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        tab = json.load(open(files[-1]))
    except Exception:
        return None, None
    e = tab.get(family)
    return (round(e["hbm_bytes_per_launch"]), os.path.basename(files[-1])) if e else (None, None)


def roofline_of(prof, steps, table=None, table_steps=0):
    """`prof`: entries recorded over the timed steps (only the dominant family when the warm-up table `table` exists);
    `table`: all families over `table_steps` warm-up steps."""
    prof = sorted(prof, key=lambda e: -e["ms"])
    d = prof[0]
    if table:
        tab = sorted(table, key=lambda e: -e["ms"])
        tot = sum(e["ms"] for e in tab) / table_steps * steps        # scaled to the timed step count
    else:
        tab, table_steps, tot = prof, steps, sum(e["ms"] for e in prof)
    m = family_math(d["name"])
    avg_ms = d["ms"] / d["launches"]
    ach = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12                    # algorithmic TFLOP/s
    ach_bw = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9                   # algorithmic GB/s
    traffic, tsrc = measured_traffic(d["name"])
    # which roof bounds this kernel: arithmetic intensity (algorithmic flop per algorithmic HBM byte) against the ridge
    # point of its MFMA peak and the 8 TB/s HBM peak
    intensity = d["flops"] / d["bytes"] if d["bytes"] else float("inf")
    ridge = PEAK_TFLOPS[m] * 1e12 / (PEAK_HBM_GBS * 1e9)
    hbm_bound = intensity < ridge
    roof = {"bound": "hbm" if hbm_bound else "mfma", "kernel": d["name"],
            "achieved": round(ach_bw if hbm_bound else ach, 2), "peak": PEAK_HBM_GBS if hbm_bound else PEAK_TFLOPS[m],
            "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": round(ach_bw / PEAK_HBM_GBS if hbm_bound else ach / PEAK_TFLOPS[m], 4),
            "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
            "flops_per_launch": d["flops"] / d["launches"],
            "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
            "avg_launch_ms": round(avg_ms, 4), "launches_per_step": round(d["launches"] / steps, 1),
            "share_of_kernel_time": round(d["ms"] / tot, 3),
            "other_roof": {"bound": "mfma" if hbm_bound else "hbm", "achieved": round(ach if hbm_bound else ach_bw, 2),
                           "peak": PEAK_TFLOPS[m] if hbm_bound else PEAK_HBM_GBS, "unit": "TFLOP/s" if hbm_bound else "GB/s",
                           "frac": round(ach / PEAK_TFLOPS[m] if hbm_bound else ach_bw / PEAK_HBM_GBS, 4)},
            "mfma_issued_per_product": MFMA_PER_PRODUCT[m],
            "frac_of_mfma_issue_peak": round(ach * MFMA_PER_PRODUCT[m] / PEAK_TFLOPS[m], 4)}
    if m in SUSTAINED_MFMA_TFLOPS:
        roof["sustained_mfma_rate"] = {"value": SUSTAINED_MFMA_TFLOPS[m], "unit": "TFLOP/s",
                                       "frac": round(ach * MFMA_PER_PRODUCT[m] / SUSTAINED_MFMA_TFLOPS[m], 4),
                                       "source": "tools/ubench/mfma_rate.hip: dense 16x16x32 MFMA stream on all CUs, random operands (profiles/r02_ubench_mfma_rate.txt)"}
    fams = [{"name": e["name"], "ms_per_step": round(e["ms"] / table_steps, 3),
             "tflops": round(e["flops"] / e["ms"] / 1e9, 2) if e["ms"] else 0.0,
             "gbs": round(e["bytes"] / e["ms"] / 1e6, 1) if e["ms"] else 0.0} for e in tab[:8]]
    roof["events"] = ("timed steps: this family only; per-family table: warm-up steps, every launch bracketed" if table
                      else "timed steps: every launch bracketed")
    return roof, fams, tot


def spawn_ranks(n: int) -> int:
    """Launch `n` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a
    child process; its stdout (rank 0's one JSON line) and stderr pass through.  Returns the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this pool's host driver
    # --standalone: torchrun's own c10d rendezvous on a port IT binds (no bind / close / reuse race with other processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without WORLD_SIZE: launching {' '.join(cmd[1:7])} ...")
    return subprocess.run(cmd, env=env).returncode


def timed_leg(eng, step, steps, warmup):
    """`step()` timed over `steps` passes (no HIP events in the timed region), then 2 passes with every launch bracketed for
    the per-family table and the dominant kernel's roofline object.  -> (seconds per step, roofline, families, last output)."""
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    eng.ctx.profile_filter(None)
    eng.ctx.profile_start(8000)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    prof = eng.ctx.profile_stop()
    roof, fams, tot = roofline_of(prof, 2)
    roof["kernel_ms_per_step"] = round(tot / 2, 3)
    # the committed PMC table (profiles/rNN_hbm_traffic.json) holds bytes per launch of the configs[1] shapes: not this leg's launches
    roof["traffic"], roof["traffic_source"] = None, "not collected for this leg (the PMC table is per launch of the configs[1] step)"
    return wall, roof, fams, out


def leg_configs4(dev, enc, voc, steps):
    """BASELINE configs[4]: blind inpainting on variable-length clips -- 32 clips per GPU, lengths U[4 s, 10 s] seed 1234 (SURVEY 8(d)
    config #5), HuBERT-base + HiFi-GAN V1.  Three routes over the same resident clips: ONE ragged batch (every launch shared, tiles
    numbered without gaps), ragged sub-batches cut at 15 % storage padding, and the exact-length route (one uniform launch set per
    clip, what `predict_ragged` did before).  RTF = true audio-seconds / wall."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, algorithmic_gmac, mel_frames
    from speech_inpainting_amd.engine import InpaintingEngine
    from speech_inpainting_amd.predict import plan_ragged_batches, storage_padding
    harch, varch = HubertArch.base(), VocoderArch.v1()
    B = 32
    g = torch.Generator().manual_seed(1234)
    secs = (4.0 + 6.0 * torch.rand(B, generator=g)).tolist()
    n16 = [int(round(s * 16000)) for s in secs]
    n22 = [-(-n * 441 // 320) for n in n16]
    w16 = [synth.synth_wave(1, n, synth.DEFAULT_SEED + 40 + i)[0] for i, n in enumerate(n16)]
    w22 = [synth.synth_wave(1, n, synth.DEFAULT_SEED + 90 + i, sr=22050)[0] for i, n in enumerate(n22)]
    eng = InpaintingEngine(harch, varch, 100, dev, enc, voc).load_state(synth.synth_hubert_state(harch), synth.synth_generator_state(varch),
                                                                       synth.synth_codebook(100))
    audio_s = sum(n / 16000.0 for n in n16)
    gflop = sum(2.0 * sum(algorithmic_gmac(harch, varch, a, mel_frames(b))) for a, b in zip(n16, n22))

    def resident(idx):
        a = torch.zeros(len(idx), max(n16[i] for i in idx))
        b = torch.zeros(len(idx), max(n22[i] for i in idx))
        for k, i in enumerate(idx):
            a[k, :n16[i]] = w16[i]
            b[k, :n22[i]] = w22[i]
        return a.to(dev), [n16[i] for i in idx], b.to(dev), [n22[i] for i in idx], torch.zeros(len(idx), dtype=torch.int32, device=dev)

    def route(plan):
        groups = [resident(idx) for idx in plan]

        def step():
            out = None
            for a, la, b, lb, pos in groups:
                mel = eng.mel_ragged(b, lb)
                out = eng.predict_ragged_batch(a, la, mel, [mel_frames(n) for n in lb], pos, 0, blind=True)
            return out
        return step

    res = {"workload": "BASELINE configs[4]: blind inpainting, 32 clips per GPU, lengths U[4 s, 10 s] (seed 1234), HuBERT-base + HiFi-GAN V1; "
                       "step = log-mel front-end -> encoder -> arg-max over ALL frames / splice -> vocoder on resident raw clips",
           "clips": B, "audio_seconds": round(audio_s, 2), "gflop_algorithmic": round(gflop, 1), "dtype": f"encoder {enc}, vocoder {voc}"}
    one = plan_ragged_batches(n16, B)
    sub = plan_ragged_batches(n16, B, 0.15)
    wall, roof, fams, out = timed_leg(eng, route(one), steps, 2)
    finite = bool(torch.isfinite(out["wave"]).all())
    res.update({"value": round(audio_s / wall, 2), "unit": "x real-time (true audio-sec/wall-sec)", "ms_per_step": round(1e3 * wall, 3), "steps": steps,
                "route": "one ragged batch of 32 (every launch shared; tiles numbered clip by clip without gaps)",
                "storage_padding": round(storage_padding(n16, one), 4), "launched_tile_padding": 0.0,
                "achieved_tflops_whole_path": round(gflop / wall / 1e3, 2), "roofline": roof, "kernel_families": fams, "finite": finite})
    wall_s, _, _, _ = timed_leg(eng, route(sub), steps, 1)
    res["ragged_subbatches"] = {"value": round(audio_s / wall_s, 2), "ms_per_step": round(1e3 * wall_s, 3), "batches": [len(b) for b in sub],
                                "storage_padding": round(storage_padding(n16, sub), 4),
                                "route": "ragged batches cut where storage padding would exceed 15 %"}
    # exact-length route: uniform entry points, one clip per launch set (continuous lengths: every bucket is a singleton)
    singles = [(w16[i][None].to(dev), w22[i][None].to(dev)) for i in range(B)]
    zero = torch.zeros(1, dtype=torch.int32, device=dev)

    def exact_step():
        out = None
        for a, b in singles:
            out = eng.predict_batch(a, eng.mel(b), zero, 0, blind=True)
        return out
    for _ in range(1):
        exact_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_exact = max(1, steps // 2)
    for _ in range(n_exact):
        exact_step()
    torch.cuda.synchronize()
    wall_e = (time.perf_counter() - t0) / n_exact
    res["exact_length_route"] = {"value": round(audio_s / wall_e, 2), "ms_per_step": round(1e3 * wall_e, 3), "launch_sets": B,
                                 "route": "exact-length buckets through the uniform entry points: 32 singletons"}
    del eng
    return res


def leg_configs3(dev, enc, voc, steps):
    """BASELINE configs[3] per GPU (batch 128 over 8 GPUs = 16 clips): (a) the I_ea path with the HuBERT-LARGE encoder and a 400 ms
    mask -- what the shipped I_ea/predict.yaml:27-28,39 selects -- and (b) I_da's inpainting() (I_da/scripts/inpainting.py:151-266):
    HuBERT-large features at layer 18 of the clean and the corrupted clips, k-means units, unit splice, F0 VQ-VAE, unit HiFi-GAN,
    both waveforms."""
    from speech_inpainting_amd import native, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, algorithmic_gmac, mel_frames
    from speech_inpainting_amd.engine import CodeGenerator, F0Quantizer, InpaintingEngine
    harch = HubertArch.large()
    B, N, K = 16, N_SAMPLES, 100
    out = {}
    hsd = synth.synth_hubert_state(harch, 77)
    # ---- (a) I_ea, large encoder, Lm = 20
    varch = VocoderArch.v1()
    eng = InpaintingEngine(harch, varch, K, dev, enc, voc).load_state(hsd, synth.synth_generator_state(varch), synth.synth_codebook(K))
    lm = 20
    T, Tm = harch.num_frames(N), mel_frames(N * 22050 // 16000)
    wave = synth.synth_wave(B, N, synth.DEFAULT_SEED + 3).to(dev)
    wave22 = synth.synth_wave(B, N * 22050 // 16000, synth.DEFAULT_SEED + 6, sr=22050).to(dev)
    pos = synth.synth_mask_frames(B, T, lm, synth.DEFAULT_SEED + 5).to(dev)
    ms, ml = (pos * 320 + 80).to(torch.int32), torch.full_like(pos, lm * 320 - 81)
    s22, e22 = (pos * 320 * 22050 // 16000).to(torch.int32), ((pos + lm) * 320 * 22050 // 16000).to(torch.int32)

    def step_iea():
        return eng.predict_batch(wave, eng.mel(wave22, s22, e22), pos, lm, mask_start=ms, mask_len=ml)
    wall, roof, fams, o = timed_leg(eng, step_iea, steps, 2)
    gflop = B * 2.0 * sum(algorithmic_gmac(harch, varch, N, Tm))
    out["configs3_iea_large"] = {
        "workload": "BASELINE configs[3] per GPU: 16 x 4 s clips, HuBERT-LARGE (24 pre-LN layers, LayerNorm feature extractor) + HiFi-GAN V1, 400 ms mask "
                    "(I_ea/predict.yaml:27-28,39); step = log-mel front-end -> encoder -> arg-max/splice -> vocoder",
        "value": round(B * CLIP_SECONDS / wall, 2), "unit": "x real-time (audio-sec/wall-sec)", "ms_per_step": round(1e3 * wall, 3), "steps": steps,
        "clips": B, "dtype": f"encoder {enc}, vocoder {voc}", "gflop_algorithmic": round(gflop, 1),
        "achieved_tflops_whole_path": round(gflop / wall / 1e3, 2), "roofline": roof, "kernel_families": fams,
        "finite": bool(torch.isfinite(o["wave"]).all())}
    del eng
    # ---- (b) I_da inpainting(), layer 18, unit HiFi-GAN (hubert_lut.json shapes)
    L = 18
    uarch = VocoderArch(upsample_rates=(5, 4, 4, 2, 2), upsample_kernel_sizes=(11, 8, 8, 4, 4), upsample_initial_channel=512, num_mels=384,
                        sampling_rate=16000)
    enc_sd = {k: v for k, v in hsd.items() if k.startswith("base_model.")}
    eng = InpaintingEngine(harch, uarch, K, dev, enc, voc).load_state(enc_sd, synth.synth_generator_state(uarch, 78))
    g = torch.Generator().manual_seed(1)
    gen = CodeGenerator(eng, torch.randn(K, 128, generator=g) * 0.5, torch.randn(20, 128, generator=g) * 0.5,
                        f0_quantizer=F0Quantizer(eng, synth.synth_f0_vqvae_state(native.F0EncDesc(), 20, seed=79)))
    wave = synth.synth_wave(B, N, 81).to(dev)
    cent = torch.randn(K, harch.hidden_size, generator=g).to(dev)
    f0 = torch.randn(B, 1, N // 80 - 3, generator=g).to(dev)
    spk = (torch.randn(B, 128, generator=g) * 0.5).to(dev)

    def step_ida():
        return eng.ida_inpaint_batch(wave, 24000, 6400, cent, gen, f0, spk, output_layer=L)
    wall, roof, fams, o = timed_leg(eng, step_ida, steps, 2)
    out["configs3_ida"] = {
        "workload": "BASELINE configs[3] per GPU, I_da's inpainting(): 16 x 4 s clips, 400 ms mask at 1.5 s, HuBERT-large features at layer 18 of the clean "
                    "AND the corrupted clips (one 32-clip encoder pass), k-means units (K = 100), unit splice, F0 VQ-VAE, unit HiFi-GAN (ups 5,4,4,2,2): "
                    "both waveforms (I_da/scripts/inpainting.py:151-266)",
        "value": round(B * CLIP_SECONDS / wall, 2), "unit": "x real-time (audio-sec of the INPUT clips/wall-sec; two waveforms per clip are produced)",
        "ms_per_step": round(1e3 * wall, 3), "steps": steps, "clips": B, "dtype": f"encoder {enc}, unit vocoder {voc}",
        "roofline": roof, "kernel_families": fams, "finite": bool(torch.isfinite(o["audio_inp"]).all() and torch.isfinite(o["audio_gen"]).all())}
    del eng
    return out


def leg_end_to_end(dev, enc, voc, steps):
    """SURVEY 8(f) row f-3, the request front (speech_inpainting_amd/stream.py): configs[1]'s clips start in HOST memory at the file's
    rate (22.05 kHz float32, what `librosa.load(sr=None)` yields) and end as int16 PCM in host memory -- pinned H2D, GPU resampling to
    16 kHz (resampy kaiser_best), masked log-mel, encoder, arg-max / splice, vocoder, int16 conversion, async D2H -- double-buffered
    over a copy stream.  RTF includes every transfer; `value` of the main line (resident tensors) is the kernel-side figure."""
    import numpy as np
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    from speech_inpainting_amd.stream import Request, RequestFront
    harch, varch = HubertArch.base(), VocoderArch.v1()
    B, n22 = 32, N_SAMPLES * 22050 // 16000
    eng = InpaintingEngine(harch, varch, 100, dev, enc, voc).load_state(synth.synth_hubert_state(harch), synth.synth_generator_state(varch),
                                                                       synth.synth_codebook(100))
    clips = list(synth.synth_wave(B, n22, synth.DEFAULT_SEED + 6, sr=22050).numpy())
    pos = synth.synth_mask_frames(B, harch.num_frames(N_SAMPLES), MASK_FRAMES, synth.DEFAULT_SEED + 5).tolist()
    rq = Request(clips, pos, MASK_FRAMES)
    front = RequestFront(eng, 22050, depth=2)
    for _ in front.run([rq] * 3):                                    # warm-up: buffers, resampler tables, kernels
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for res in front.run([rq] * steps):
        n += len(res.pcm)
    wall = time.perf_counter() - t0
    t1 = time.perf_counter()
    for res in front.run([rq]):
        pass
    lat = time.perf_counter() - t1
    bytes_in, bytes_out = B * n22 * 4, sum(len(p) for p in res.pcm) * 2
    del eng
    return {"workload": "BASELINE configs[1] from HOST clips to HOST PCM: 32 x 4 s clips at 22.05 kHz float32 in host memory -> pinned H2D -> GPU resample "
                        "(resampy kaiser_best) -> masked log-mel -> encoder -> arg-max/splice -> vocoder -> int16 on the GPU -> async D2H; batches double-buffered",
            "value": round(n * CLIP_SECONDS / wall, 2), "unit": "x real-time (audio-sec/wall-sec), PCIe both ways included", "ms_per_step": round(1e3 * wall / steps, 3),
            "steps": steps, "single_batch_latency_ms": round(1e3 * lat, 3), "h2d_bytes_per_step": bytes_in, "d2h_bytes_per_step": bytes_out,
            "dtype": f"encoder {enc}, vocoder {voc}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--encoder-dtype", default="bf16", choices=["fp32", "bf16", "bf16x3"])
    ap.add_argument("--vocoder-dtype", default="fp16", choices=["fp32", "bf16", "bf16x3", "fp16"])
    ap.add_argument("--vocoder-chunk", type=int, default=0)
    ap.add_argument("--cpu-clips", type=int, default=8, help="clips timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the secondary exact-fp32-vocoder leg")
    ap.add_argument("--graph-leg", action="store_true", help="also replay the step as one hipGraph (informational)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the configs[3] / configs[4] legs (extra keys of the JSON line)")
    ap.add_argument("--only-config-legs", action="store_true", help="diagnostic: run ONLY the configs[3] / configs[4] legs and print them")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes and relay rank 0's JSON
        # line.  Done before anything here touches the GPU (torch.cuda.device_count() does not initialise it), and as a
        # child process, never an exec of this one.
        raise SystemExit(spawn_ranks(a.gpus))

    from speech_inpainting_amd import parallel, synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames
    from speech_inpainting_amd.engine import InpaintingEngine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    backend = os.environ.get("SI_DIST_BACKEND", "nccl")      # "gloo" only to rehearse N > 1 on a single GPU
    rank, local_rank, world = parallel.init_distributed(backend)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", local_rank if local_rank < torch.cuda.device_count() else 0)
    torch.cuda.set_device(dev)

    if a.only_config_legs:
        if rank == 0:
            r = {"configs4": leg_configs4(dev, a.encoder_dtype, a.vocoder_dtype, max(3, a.steps // 2)),
                 "end_to_end": leg_end_to_end(dev, a.encoder_dtype, a.vocoder_dtype, max(4, a.steps))}
            r.update(leg_configs3(dev, a.encoder_dtype, a.vocoder_dtype, max(3, a.steps // 2)))
            print(json.dumps(r), flush=True)
        return

    harch, varch = HubertArch.base(), VocoderArch.v1()
    K = 100
    B = a.batch
    T = harch.num_frames(N_SAMPLES)
    Tm = mel_frames(N_SAMPLES * 22050 // 16000)
    state = {}

    def checkpoint():
        if not state:
            state["sd"] = (synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(K))
        return state["sd"]

    # this rank's slice of the global utterance list (seeded per global clip index)
    lo, hi = parallel.shard_range(B * world, rank, world)
    wave = synth.synth_wave(hi - lo, N_SAMPLES, synth.DEFAULT_SEED + 3 + lo).to(dev)
    wave22 = synth.synth_wave(hi - lo, N_SAMPLES * 22050 // 16000, synth.DEFAULT_SEED + 6 + lo, sr=22050).to(dev)
    pos = synth.synth_mask_frames(hi - lo, T, MASK_FRAMES, synth.DEFAULT_SEED + 5 + lo).to(dev)
    mstart = (pos * 320 + 80).to(torch.int32)
    mlen = torch.full_like(pos, MASK_FRAMES * 320 - 81)
    s22 = (pos * 320 * 22050 // 16000).to(torch.int32)                       # I_ea/predict.py:99-100
    e22 = ((pos + MASK_FRAMES) * 320 * 22050 // 16000).to(torch.int32)

    def run_mode(enc, voc, steps, warmup, events, graph_leg=False):
        t_load = time.perf_counter()
        eng = parallel.setup_engine(lambda: InpaintingEngine(harch, varch, K, dev, enc, voc, a.vocoder_chunk), checkpoint, rank)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] encoder {enc}, vocoder {voc}: setup {time.perf_counter() - t_load:.1f} s; B={B}/GPU x {world} GPU, T={T}, Tm={Tm}")

        def step():
            # raw 16 kHz + 22.05 kHz clips in HBM -> masked log-mel (f-1) -> encoder -> arg-max/splice -> vocoder
            mel = eng.mel(wave22, s22, e22)
            return eng.predict_batch(wave, mel, pos, MASK_FRAMES, mask_start=mstart, mask_len=mlen)

        # Warm-up: every launch bracketed by HIP events (all but the first warm-up step) -> the per-family table and the
        # dominant family.  Timed steps: only the dominant family is bracketed, because an event pair per launch costs
        # ~1.5 ms of a 24 ms step (measured: 23.8 vs 22.3 ms); its average launch time is what `roofline` uses.
        prof_warm, warm_steps = [], 0
        for w in range(warmup):
            if events and w == min(1, warmup - 1):
                torch.cuda.synchronize()
                eng.ctx.profile_filter(None)
                eng.ctx.profile_start(1200 * warmup)
            out = step()
            if events and w >= min(1, warmup - 1):
                warm_steps += 1
        torch.cuda.synchronize()
        if events and warm_steps:
            prof_warm = eng.ctx.profile_stop()
        if events:
            if prof_warm:
                eng.ctx.profile_filter(max(prof_warm, key=lambda e: e["ms"])["name"])
            eng.ctx.profile_start(1200 * max(steps, 1))
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = time.perf_counter() - t0
        prof = eng.ctx.profile_stop() if events else []
        # extra, informational: the same step replayed as ONE hipGraph (no per-launch host work, no HIP events) -- what the
        # ~230 dependent launches of a step cost in gaps.  Never `value`: the contract's roofline needs events in the timed region.
        graph_ms = None
        if graph_leg and world == 1:
            try:
                eng.ctx.profile_filter(None)
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()
                torch.cuda.current_stream().wait_stream(side)
                with torch.cuda.graph(g):
                    gout = step()
                g.replay(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    g.replay()
                torch.cuda.synchronize()
                graph_ms = 1e3 * (time.perf_counter() - t1) / steps
                if not torch.equal(gout["labels"], out["labels"]):
                    graph_ms = None
                del g
            except Exception as ex:      # capture is best-effort; the measured legs above do not depend on it
                log(f"[bench] hipGraph leg skipped: {ex!r}")
        wav = out["wave"]
        finite = bool(torch.isfinite(wav).all())
        stats = parallel.gather_metrics([elapsed, float(hi - lo), float(wav.pow(2).mean().sqrt()), float(finite)],
                                        dev if backend == "nccl" else "cpu").cpu()
        del eng
        elapsed_max = float(stats[:, 0].max())
        clips = float(stats[:, 1].sum())
        clips_slowest = float(stats[int(stats[:, 0].argmax()), 1])           # the clips of the rank that set the wall time
        if not bool(stats[:, 3].min()):
            raise SystemExit("non-finite samples in the output waveform")
        return dict(elapsed=elapsed_max, clips=clips, clips_slowest_rank=clips_slowest, rms=float(stats[0, 2]), prof=prof, steps=steps, prof_warm=prof_warm, graph_ms=graph_ms,
                    warm_steps=warm_steps,
                    wave=wav if rank == 0 else None, labels=out["labels"] if rank == 0 else None)

    events = not a.no_kernel_events
    main_run = run_mode(a.encoder_dtype, a.vocoder_dtype, a.steps, a.warmup, events, graph_leg=a.graph_leg)
    # reference legs on the same inputs: the fp32-equivalent split mode and exact fp32 (fewer steps; reported, not `value`)
    legs = {}
    if not a.no_fp32_leg:
        for voc in ("bf16x3", "fp32"):
            if voc != a.vocoder_dtype:
                legs[voc] = run_mode(a.encoder_dtype, voc, max(2, a.steps // 3), 2, events)
    if rank != 0:
        return

    def headline(r):
        return r["clips"] * r["steps"] * CLIP_SECONDS / r["elapsed"]

    dtype_txt = {"fp32": "fp32 (exact fp32 MFMA)", "bf16": "bf16 MFMA, fp32 accumulate",
                 "bf16x3": "bf16x3 (fp32 operands split hi+lo, 3 bf16 MFMAs per product, fp32 accumulate: fp32-equivalent)",
                 "fp16": "fp16 MFMA (operands rounded to fp16, saturating), fp32 accumulate, activations stored as fp16"}
    clips = main_run["clips"]
    res = {
        "metric": "real-time factor (audio-sec/wall-sec) per GPU, 4 s clips @16 kHz, 200 ms mask",
        "value": round(headline(main_run), 2), "unit": "x real-time (audio-sec/wall-sec); `value` = whole-job aggregate over n_gpus, `value_per_gpu` = the per-GPU figure the metric names",
        "value_per_gpu": round(main_run["clips_slowest_rank"] * main_run["steps"] * CLIP_SECONDS / main_run["elapsed"], 2),
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * main_run["elapsed"] / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"encoder GEMMs and attention products: {dtype_txt[a.encoder_dtype]} (softmax, LayerNorm, head, arg-max in fp32); vocoder: {dtype_txt[a.vocoder_dtype]}",
        "data": "synthetic (seeded clips, random-init weights of the HuBERT-base + HiFi-GAN V1 architecture)",
        "config": {"workload": "BASELINE configs[1]: batch=32 x 4 s clips per GPU, 200 ms mask, HuBERT-base + HiFi-GAN V1; step = masked log-mel front-end -> encoder -> arg-max/splice -> vocoder on resident raw clips"
                               + ("" if world == 1 else f", utterance-sharded over {world} GPUs (configs[2] at 8)"),
                   "global_batch": int(clips), "clip_samples": N_SAMPLES, "mask_frames": MASK_FRAMES,
                   "parallelism": f"utterance-sharded x{world}", "output_rms": round(main_run["rms"], 4)},
        "clips_per_s": round(clips * a.steps / main_run["elapsed"], 2),
        "gflop_per_clip_algorithmic": GFLOP_PER_CLIP,
        "achieved_tflops_whole_path": round(GFLOP_PER_CLIP * 1e9 * clips * a.steps / main_run["elapsed"] / 1e12, 2),
    }

    def table(r, tag):
        roof, fams, tot = roofline_of(r["prof"], r["steps"], r["prof_warm"], r["warm_steps"])
        rows, nst = (r["prof_warm"], r["warm_steps"]) if r["prof_warm"] else (r["prof"], r["steps"])
        log(f"[bench] {tag}: per-kernel HIP-event time (rank 0, {nst} {'warm-up' if r['prof_warm'] else 'timed'} steps): "
            f"{tot / r['steps']:.2f} ms/step in kernels; timed: {1e3 * r['elapsed'] / r['steps']:.2f} ms/step wall, "
            f"{roof['kernel']} {roof['avg_launch_ms']:.4f} ms/launch")
        for e in sorted(rows, key=lambda e: -e["ms"]):
            log(f"    {e['name']:<28} {e['launches'] / nst:7.1f} launches/step {e['ms'] / nst:9.3f} ms/step "
                f"{e['flops'] / e['ms'] / 1e9 if e['ms'] else 0:8.2f} TFLOP/s {e['bytes'] / e['ms'] / 1e6 if e['ms'] else 0:9.1f} GB/s (algorithmic)")
        return roof, fams

    if main_run["prof"]:
        res["roofline"], res["kernel_families"] = table(main_run, f"{a.encoder_dtype}/{a.vocoder_dtype}")
    if "bf16x3" in legs:
        # the number of record at the REFERENCE's vocoder precision (fp32-equivalent operands), next to `value`
        res["reference_precision_value"] = round(headline(legs["bf16x3"]), 2)
        res["reference_precision_note"] = ("same step with the vocoder on bf16x3 (every fp32 operand split hi + lo, fp32 accumulate, fp32 "
                                           "activations: 1.5e-6 waveform RMS vs the fp32 reference); `value` is the fp16-MFMA vocoder "
                                           "(1.35e-4 RMS, inside the 1e-3 gate)")
    for voc, r in legs.items():
        leg = {"value": round(headline(r), 2), "ms_per_step": round(1e3 * r["elapsed"] / r["steps"], 3),
               "steps": r["steps"], "dtype": f"encoder {a.encoder_dtype}, vocoder {dtype_txt[voc]}"}
        if r["prof"]:
            leg["roofline"], leg["kernel_families"] = table(r, f"{a.encoder_dtype}/{voc}")
        res[f"{voc}_vocoder"] = leg
    if main_run.get("graph_ms"):
        res["graph_replay"] = {"ms_per_step": round(main_run["graph_ms"], 3),
                               "value": round(clips * CLIP_SECONDS / (main_run["graph_ms"] * 1e-3), 2),
                               "note": "the same step captured once and replayed as one hipGraph: no per-launch host work and no "
                                       "HIP events; informational, not the headline"}
    if "fp32" in legs:
        # same clips, same encoder arithmetic: the difference is the vocoder's operand rounding alone
        ref, got = legs["fp32"]["wave"], main_run["wave"]
        res["vocoder_check"] = {
            "waveform_rms_vs_fp32_vocoder_leg": float((got - ref).pow(2).mean().sqrt()),
            "fp32_leg_waveform_rms": float(ref.pow(2).mean().sqrt()),
            "labels_identical": bool(torch.equal(main_run["labels"], legs["fp32"]["labels"])),
            "gate": "north star: waveform RMS error <= 1e-3 (fp32 waveform)"}
    if world == 1 and not a.no_config_legs:
        # the other single-GPU configurations of BASELINE.json, each with its own roofline object; `value` above stays configs[1]
        try:
            res["configs4"] = leg_configs4(dev, a.encoder_dtype, a.vocoder_dtype, max(3, a.steps // 2))
            log(f"[bench] configs4 (32 ragged clips, blind): {res['configs4']['value']} x RT, {res['configs4']['ms_per_step']} ms/step; "
                f"sub-batches {res['configs4']['ragged_subbatches']['value']}, exact-length route {res['configs4']['exact_length_route']['value']}")
            res["end_to_end"] = leg_end_to_end(dev, a.encoder_dtype, a.vocoder_dtype, max(4, a.steps))
            log(f"[bench] end_to_end (host clips -> host PCM, double-buffered): {res['end_to_end']['value']} x RT, {res['end_to_end']['ms_per_step']} ms/step, "
                f"single-batch latency {res['end_to_end']['single_batch_latency_ms']} ms")
            res.update(leg_configs3(dev, a.encoder_dtype, a.vocoder_dtype, max(3, a.steps // 2)))
            log(f"[bench] configs3_iea_large: {res['configs3_iea_large']['value']} x RT ({res['configs3_iea_large']['ms_per_step']} ms); "
                f"configs3_ida: {res['configs3_ida']['value']} x RT ({res['configs3_ida']['ms_per_step']} ms)")
        except Exception as ex:
            res["config_legs_error"] = repr(ex)
            log(f"[bench] configs[3] / configs[4] legs failed: {ex!r}")
    if world == 1 and a.cpu_clips > 0:
        try:
            res["cpu_baseline"] = cpu_baseline(a.cpu_clips)
        except Exception as ex:     # the oracle is a checker; its absence must not void the GPU number
            res["cpu_baseline"] = {"error": repr(ex)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
