"""GPU parity of the mel front-end: si_mel_frontend through the C ABI vs the oracle's
restatement of I_ea/predict.py:99-106 + I_ea/dataset/mel_dump.py:40-98.

Tolerance: the reference computes the STFT with an fp32 FFT, the HIP path as an exact-fp32 DFT GEMM; both carry
~1e-6 relative rounding on the magnitudes, and log() turns a relative error into an absolute one, so the gate is an
absolute 2e-4 on the log-mel (values span [-11.5, +3]); typical error is ~1e-5.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from speech_inpainting_amd import synth
from speech_inpainting_amd.arch import HubertArch, VocoderArch, mel_frames

pytestmark = pytest.mark.gpu

MEL_ATOL = 2e-4


@pytest.fixture(scope="module")
def ctx():
    from speech_inpainting_amd import native
    c = native.NativeContext(native.make_desc(HubertArch.tiny(), VocoderArch.tiny(), 10), torch.device("cuda:0"))
    yield c
    c.close()


def _clips(B, n, seed):
    return synth.synth_wave(B, n, seed, sr=22050).numpy()


@pytest.mark.parametrize("n22", [88200, 66150, 22063, 220500])
def test_masked_mel_matches_oracle(ctx, n22):
    B = 3
    w = _clips(B, n22, 11) * np.array([[0.3], [1.7], [0.05]], dtype=np.float32)       # peaks below and above 1
    starts = [n22 // 3, 0, n22 - 2000]
    ends = [n22 // 3 + 4410, 1500, n22]                                               # interior, at the head, at the tail
    ref = R.masked_mel(w, starts, ends)
    dev = torch.device("cuda:0")
    got = ctx.mel_frontend(torch.from_numpy(w).to(dev), torch.tensor(starts, dtype=torch.int32, device=dev),
                           torch.tensor(ends, dtype=torch.int32, device=dev))
    assert got.shape == ref.shape == (B, 80, mel_frames(n22))
    err = (got.cpu() - ref).abs()
    assert float(err.max()) <= MEL_ATOL, float(err.max())
    assert float(err.mean()) <= 2e-5, float(err.mean())


def test_get_mel_without_mask_or_normalisation(ctx):
    w = _clips(2, 44100, 5) * 0.4
    ref = R.mel_spectrogram(torch.from_numpy(w))
    got = ctx.mel_frontend(torch.from_numpy(w).cuda(), None, None, normalize=False)
    assert float((got.cpu() - ref).abs().max()) <= MEL_ATOL
    # normalised, no mask (the `orig` figure of predict.py:92-95)
    ref2 = R.masked_mel(w, None, None)
    got2 = ctx.mel_frontend(torch.from_numpy(w).cuda())
    assert float((got2.cpu() - ref2).abs().max()) <= MEL_ATOL


def test_silent_and_fully_masked_clips(ctx):
    n = 30000
    w = _clips(2, n, 3)
    w[0] = 0.0
    s, e = torch.tensor([0, 0], dtype=torch.int32).cuda(), torch.tensor([0, n], dtype=torch.int32).cuda()
    got = ctx.mel_frontend(torch.from_numpy(w).cuda(), s, e).cpu()
    ref = R.masked_mel(w, [0, 0], [0, n])
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 1e-5          # sqrt(1e-9) floor through the mel basis
    assert torch.equal(got[0], got[1])


def test_peak_normalisation_makes_gain_irrelevant_at_full_batch(ctx):
    """Size-independent property on the bench-size batch: power-of-two gains are exact in fp32, so the normalised
    mel of 4x and x/8 must be bit-identical to that of x."""
    B, n = 32, 88200
    w = torch.from_numpy(_clips(B, n, 17)).cuda()
    s = torch.full((B,), 40000, dtype=torch.int32, device="cuda")
    e = s + 4410
    a = ctx.mel_frontend(w, s, e)
    b = ctx.mel_frontend(w * 4.0, s, e)
    c = ctx.mel_frontend(w * 0.125, s, e)
    assert torch.equal(a, b) and torch.equal(a, c)
    # and the masked span shows up as the floor value in the frames that lie wholly inside it
    inside = a[:, :, (40000 + 312) // 441 + 3:(44410 + 312 - 1024) // 441 - 1]
    assert inside.numel() > 0 and float(inside.max()) < -9.0


def test_too_short_clip_is_refused(ctx):
    from speech_inpainting_amd.native import NativeError
    with pytest.raises((NativeError, ValueError)):
        ctx.mel_frontend(torch.zeros(1, 300, device="cuda"))


def test_kmeans_assignment_matches_sklearn_and_oracle(ctx):
    """f-2: si_kmeans_assign vs sklearn's KMeans.predict (the call the reference makes, I_da/scripts/inpainting.py:204-205)
    and vs the oracle's restatement of I_ea/dataset/km_label.py:20-24.  Labels must agree wherever the two best
    centroids are not within fp32 rounding of each other."""
    from sklearn.cluster import KMeans
    g = torch.Generator().manual_seed(9)
    for rows, D, K in ((6368, 768, 100), (500, 80, 500), (7, 1024, 3)):
        cent = torch.randn(K, D, generator=g)
        lab_true = torch.randint(0, K, (rows,), generator=g)
        x = cent[lab_true] + 0.9 * torch.randn(rows, D, generator=g)      # noisy members of known clusters
        km = KMeans(n_clusters=K, n_init=1)
        km.cluster_centers_ = cent.numpy().astype(np.float32)
        km._n_threads = 1
        ref_sk = torch.from_numpy(km.predict(x.numpy().astype(np.float32))).long()
        ref_or = R.kmeans_assign(x, cent)
        got, dist = ctx.kmeans_assign(x.cuda(), cent.cuda(), with_distance=True)
        got, dist = got.cpu(), dist.cpu()
        d_all = torch.cdist(x.double(), cent.double()).pow(2)
        top2 = d_all.topk(2, dim=1, largest=False).values
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * top2[:, 1]
        assert clear.float().mean() > 0.99
        assert torch.equal(got[clear], ref_sk[clear]) and torch.equal(got[clear], ref_or[clear]), (rows, D, K)
        assert torch.equal(got[clear], d_all.argmin(1)[clear])
        assert torch.allclose(dist.double(), top2[:, 0], rtol=1e-4, atol=1e-3)


def test_gpu_resampler_matches_scipy_resample_poly(ctx):
    """f-3: si_resample_poly vs scipy.signal.resample_poly with the same Kaiser filter (float64 reference), for the two
    conversions of the predict path (22.05 kHz file -> 16 kHz; 16 kHz file -> 22.05 kHz) and an odd length."""
    from scipy.signal import resample_poly
    from speech_inpainting_amd import audio
    for sr_in, sr_out, n in ((22050, 16000, 66150), (16000, 22050, 48000), (22050, 16000, 12347), (44100, 16000, 30000)):
        x = _clips(2, n, 23)
        taps, up, down, pre, n_out = audio.design_resampler(sr_in, sr_out, n)
        got = ctx.resample_poly(torch.from_numpy(x).cuda(), torch.from_numpy(taps).cuda(), up, down, pre, n_out).cpu().numpy()
        ref = resample_poly(x.astype(np.float64), up, down, axis=1, window=("kaiser", audio.KAISER_BETA))
        assert got.shape == ref.shape == (2, n_out)
        err = np.abs(got - ref).max()
        assert err <= 2e-6 * max(1.0, np.abs(ref).max()), (sr_in, sr_out, n, err)
        assert np.allclose(audio.resample(x[0], sr_in, sr_out), got[0], atol=3e-6)          # the host helper agrees too


def test_gpu_kaiser_best_resampler_reproduces_the_reference_held_16k_file(ctx):
    """f-3, pinned: si_resample_sinc (resampy's `kaiser_best`, librosa 0.9.1's resampler at I_ea/predict.py:79-80) on the 22.05 kHz
    samples of the reference-held LJ001-0001 pair must give the 16 kHz FILE's samples after its int16 quantisation -- the start of
    the file bit for bit (an excerpt from sample 0 shares the whole file's time registers), and agrees with the oracle (which the
    CPU test pins on all three excerpts) to 1e-7 everywhere, ragged batches included."""
    import json
    import os
    from oracle import ref_cpu as R
    from speech_inpainting_amd import audio
    from tests.common import GOLDEN
    z = np.load(os.path.join(GOLDEN, "lj001_resample.npz"))
    x = z["head22"].astype(np.float32) / 32768.0
    f = audio.design_kaiser_best(22050, 16000, len(x))
    dev = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    got = ctx.resample_sinc(torch.from_numpy(x)[None].cuda(), dev, f["n_out"]).cpu().numpy()[0]
    n = int(len(x) * 16000 / 22050)
    assert np.array_equal(np.floor(got[:n - 80].astype(np.float64) * 32768.0), z["head16"][:n - 80].astype(np.float64))
    assert got.shape == (f["n_out"],) and not got[n:].any()
    # against the oracle on other lengths / ratios, and a ragged batch against its clips alone
    for sr_in, sr_out, n_in in ((22050, 16000, 12347), (16000, 22050, 9000), (44100, 16000, 30000)):
        xs = _clips(3, n_in, 29)
        f = audio.design_kaiser_best(sr_in, sr_out, n_in)
        dev = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        lens = [n_in, n_in - 1234, n_in // 2]
        got = ctx.resample_sinc(torch.from_numpy(xs).cuda(), dev, f["n_out"], torch.tensor(lens, dtype=torch.int32).cuda()).cpu().numpy()
        for b, nb in enumerate(lens):
            ref = R.resample_kaiser_best(xs[b, :nb], sr_in, sr_out)
            assert np.abs(got[b, :len(ref)] - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (sr_in, sr_out, b)
            assert not got[b, len(ref):].any()


def test_gpu_pcm16_is_the_scripts_truncating_cast(ctx):
    """B6 on the device (si_pcm16) against the script's own two statements `audio * 32768` + `.astype('int16')`
    (I_ea/predict.py:204-206) on every value where that cast is defined, 32767 at an exactly saturated +1.0."""
    g = torch.Generator().manual_seed(3)
    a = torch.cat([torch.tensor([0.99999, -0.99999, 1.5 / 32768, -1.5 / 32768, 1.0, -1.0, 0.5, -0.25, 0.0, -0.0]),
                   torch.tanh(3.0 * torch.randn(100003, generator=g))])
    got = ctx.pcm16(a.cuda().contiguous()).cpu().numpy()
    with np.errstate(invalid="ignore"):
        ref = (a * 32768).numpy().astype("int16")
    ok = (a * 32768).numpy() < 32768.0
    assert np.array_equal(got[ok], ref[ok]) and (got[~ok] == 32767).all() and int((~ok).sum()) >= 1
    from speech_inpainting_amd import audio
    assert np.array_equal(got, audio.to_int16_pcm(a))


def test_kmeans_mfma_kernel_agrees_with_the_scalar_kernel_up_to_near_ties(ctx):
    """The k-means assignment on the matrix pipe (exact-fp32 MFMA, the D reduction split over four waves) against the one-row-per-
    workgroup scalar kernel (SI_KMEANS_MFMA=0) and the float64 distances: identical labels wherever the two best distances differ by
    more than 1e-4 relative, squared distances within 1e-4; shapes of the I_da call (6368 x 1024, K = 100), the F0 bottleneck
    (800 x 128, K = 20), K > 128 (two centroid blocks), and a ragged last row tile."""
    import os
    g = torch.Generator().manual_seed(41)
    for rows, D, K in ((6368, 1024, 100), (800, 128, 20), (1001, 256, 300), (33, 64, 7)):
        cent = torch.randn(K, D, generator=g)
        x = cent[torch.randint(0, K, (rows,), generator=g)] + 0.7 * torch.randn(rows, D, generator=g)
        got, dist = ctx.kmeans_assign(x.cuda(), cent.cuda(), with_distance=True)
        os.environ["SI_KMEANS_MFMA"] = "0"
        try:
            old, dist_old = ctx.kmeans_assign(x.cuda(), cent.cuda(), with_distance=True)
        finally:
            os.environ.pop("SI_KMEANS_MFMA", None)
        d_all = torch.cdist(x.double(), cent.double()).pow(2)
        top2 = d_all.topk(2, dim=1, largest=False).values
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * top2[:, 1]
        assert clear.float().mean() > 0.98
        got, old = got.cpu(), old.cpu()
        assert torch.equal(got[clear], d_all.argmin(1)[clear]) and torch.equal(old[clear], got[clear]), (rows, D, K)
        assert int(got.min()) >= 0 and int(got.max()) < K
        assert torch.allclose(dist.cpu().double(), top2[:, 0], rtol=1e-4, atol=1e-3) and torch.allclose(dist_old.cpu().double(), top2[:, 0], rtol=1e-4, atol=1e-3)
