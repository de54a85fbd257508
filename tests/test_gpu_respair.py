"""The fused ResBlock-pair kernels (respair.hip: C = 32 / 64; respair_wide.hip: C = 128 / 256) against the two-launch
tap-GEMM form of the same arithmetic (SI_VOC_FUSE=0) and against the fp32 oracle, on the V1 generator (all four widths)
at clip lengths that exercise interior tiles, ragged last tiles and clips shorter than one tile."""
import os

import pytest
import torch

from tests.common import rms

pytestmark = pytest.mark.gpu


def _engine(varch, gsd, fuse, voc="fp16", chain=True):
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch = HubertArch.tiny()
    fuse = fuse if isinstance(fuse, str) else ("1" if fuse else "0")   # a string: SI_VOC_FUSE as a mask
    want = {"SI_VOC_FUSE": fuse, "SI_VOC_CHAIN": "1" if chain else "0"}   # read when the context is created
    old = {k: os.environ.get(k) for k in want}
    os.environ.update(want)
    try:
        eng = InpaintingEngine(harch, varch, 20, "cuda:0", "fp32", voc)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return eng.load_state(synth.synth_hubert_state(harch), gsd, synth.synth_codebook(20))


@pytest.mark.parametrize("B,Tm", [(3, 57), (2, 1), (1, 130), (5, 9)])
def test_resblock_chain_kernel_is_bit_identical_to_the_pair_kernels(B, Tm):
    """reschain.hip (the C = 32 stage's resblock as one kernel) does the pair kernels' arithmetic in the same order: the
    waveforms must be EQUAL, on clips of several tiles (57 / 130 frames: 14592 / 33280 rows against 648-744 stored rows per
    tile), clips shorter than one tile (1 frame = 256 rows) and batches that leave workgroups with different tile counts."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch.v1()
    gsd = synth.synth_generator_state(varch)
    mel = synth.synth_mel(B, Tm, 80, 78)
    chain = _engine(varch, gsd, True, chain=True).vocode(mel.cuda(), stretch=False).cpu()
    pairs = _engine(varch, gsd, True, chain=False).vocode(mel.cuda(), stretch=False).cpu()
    assert chain.shape == pairs.shape == (B, Tm * 256)
    assert bool(torch.isfinite(chain).all())
    d = (chain - pairs).abs().max().item()
    print(f"B={B} Tm={Tm}: max |chain - pairs| = {d:.3e}, signal rms {rms(pairs):.3f}")
    assert torch.equal(chain, pairs)


@pytest.mark.parametrize("B,Tm", [(3, 57), (1, 1), (2, 130), (5, 3)])
def test_streaming_upsamplers_match_the_tap_gemm_form_and_oracle(B, Tm):
    """upsample.hip (the 128- and 64-channel transposed convolutions as persistent streaming GEMMs) against the tap-GEMM on the
    same fp16 operands and against the fp32 oracle: SI_VOC_FUSE=480 keeps every ResBlock kernel (mask bits 32 | 64 | 128 |
    256) and clears bit 1, the streaming upsamplers.  The two forms sum the same products in fp32 in a different order and
    round to fp16 once; a rounding that flips is amplified by the layers behind it until the outputs are about as far apart
    as either is from fp32 (the same bound as for the fused ResBlock kernels below).  Clips of many tiles (130 frames: 22 /
    44 tiles per clip), ragged last tiles, one-frame clips (shorter than a tile), batches that leave workgroups with
    different tile counts; and the clamp at +-65504 must act the same."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch.v1()
    gsd = synth.synth_generator_state(varch)
    mel = synth.synth_mel(B, Tm, 80, 81)
    ref = R.generator_forward(gsd, varch, mel)[:, 0, :]
    new = _engine(varch, gsd, True).vocode(mel.cuda(), stretch=False).cpu()
    old = _engine(varch, gsd, "480").vocode(mel.cuda(), stretch=False).cpu()
    assert new.shape == old.shape == ref.shape == (B, Tm * 256) and bool(torch.isfinite(new).all())
    e_n, e_o, e_no = rms(new, ref), rms(old, ref), rms(new, old)
    print(f"B={B} Tm={Tm}: signal rms {rms(ref):.3f}; streaming upsamplers vs oracle {e_n:.3e}, tap-GEMM form vs oracle {e_o:.3e}, one vs the other {e_no:.3e}")
    assert e_n <= 2e-4 and e_o <= 2e-4 and e_no <= 2e-4
    assert e_n <= 1.5 * e_o + 1e-5                                                    # no worse than the form it replaces
    hot = synth.synth_mel(2, 9, 80, 79) * 3.0e4                                       # drives the stream into fp16 overflow
    a = _engine(varch, gsd, True).vocode(hot.cuda(), stretch=False).cpu()
    c = _engine(varch, gsd, "480").vocode(hot.cuda(), stretch=False).cpu()
    assert bool(torch.isfinite(a).all()) and ((a > 0) == (c > 0)).float().mean().item() > 0.9


@pytest.mark.parametrize("B,Tm", [(3, 57), (1, 1), (2, 130), (32, 200), (5, 3)])
def test_early_upsamplers_on_gemmcu_match_the_tap_gemm_form_and_oracle(B, Tm):
    """gemmcu.hip's TC instantiations (the 512 -> 256 and 256 -> 128 channel transposed convolutions as two-tap GEMMs with N = 8 Cout:
    per-clip descriptors whose range check supplies the zero rows, the leaky-ReLU moved into the producers' epilogues, the output row
    shifted by pad * Cout and cropped) against the tap-GEMM on the same fp16 operands (SI_VOC_UPSGEMM=0) and against the fp32 oracle.  The two forms sum the
    same products in fp32 in a different order and round to fp16 once; a rounding that flips is amplified by the layers behind it
    (the bound of the streaming upsamplers' test above).  One-frame clips (the whole clip is padding rows of one tile), clips of
    several row blocks with a ragged last one, a batch that is several rounds of tiles; every frame at a clip's edge depends on the
    zero rows; and a stream driven into fp16 overflow must saturate the same way."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch.v1()
    gsd = synth.synth_generator_state(varch)
    mel = synth.synth_mel(B, Tm, 80, 83)
    ref = R.generator_forward(gsd, varch, mel[:4])[:, 0, :]

    def run(flag, m):
        os.environ["SI_VOC_UPSGEMM"] = flag
        try:
            eng = _engine(varch, gsd, True)
        finally:
            os.environ.pop("SI_VOC_UPSGEMM", None)
        eng.ctx.profile_start(4000)
        w = eng.vocode(m.cuda(), stretch=False).cpu()
        names = {e["name"] for e in eng.ctx.profile_stop()}
        assert any(n.startswith("gemmcu_f16_") for n in names) == (flag != "0"), names
        return w

    new, old = run("1", mel), run("0", mel)
    assert new.shape == old.shape == (B, Tm * 256) and bool(torch.isfinite(new).all())
    assert torch.equal(run("1", mel), new)                                            # run to run
    e_n, e_o, e_no = rms(new[:4], ref), rms(old[:4], ref), rms(new, old)
    edge = max(rms(new[:, :2048], old[:, :2048]), rms(new[:, -2048:], old[:, -2048:]))
    print(f"B={B} Tm={Tm}: signal rms {rms(ref):.3f}; gemmcu upsamplers vs oracle {e_n:.3e}, tap-GEMM form vs oracle {e_o:.3e}, one vs the other {e_no:.3e} "
          f"(clip edges {edge:.3e})")
    assert e_n <= 2e-4 and e_o <= 2e-4 and e_no <= 2e-4 and edge <= 4e-4
    assert e_n <= 1.5 * e_o + 1e-5                                                    # no worse than the form it replaces
    hot = synth.synth_mel(2, 9, 80, 79) * 3.0e4                                       # drives the stream into fp16 overflow
    a, c = run("1", hot), run("0", hot)
    assert bool(torch.isfinite(a).all()) and ((a > 0) == (c > 0)).float().mean().item() > 0.9
    if B > 1:                                                                         # a clip does not depend on its batch neighbours
        os.environ["SI_VOC_UPSGEMM"] = "1"
        try:
            eng = _engine(varch, gsd, True)
        finally:
            os.environ.pop("SI_VOC_UPSGEMM", None)
        assert torch.equal(eng.vocode(mel[1:2].cuda(), stretch=False).cpu(), new[1:2])


def test_resblock_chain_kernel_saturates_like_the_pair_kernels():
    """The tap-GEMM form clamps to +-65504 before every fp16 rounding; the fused kernels set MODE.FP16_OVFL instead, so
    that the conversion itself saturates.  A mel scaled until the activation stream overflows fp16 must give a finite
    waveform (an inf anywhere would turn into NaN one convolution later), the same one through the chain and the pair
    kernels, and close to the clamping form's."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch.v1()
    gsd = synth.synth_generator_state(varch)
    mel = synth.synth_mel(2, 9, 80, 79) * 3.0e4
    chain = _engine(varch, gsd, True, chain=True).vocode(mel.cuda(), stretch=False).cpu()
    pairs = _engine(varch, gsd, True, chain=False).vocode(mel.cuda(), stretch=False).cpu()
    plain = _engine(varch, gsd, False).vocode(mel.cuda(), stretch=False).cpu()      # tap-GEMM form: explicit clamps
    assert bool(torch.isfinite(chain).all()) and bool(torch.isfinite(pairs).all()) and bool(torch.isfinite(plain).all())
    agree = ((chain > 0) == (plain > 0)).float().mean().item()
    print(f"saturated run: |wave| mean {chain.abs().mean().item():.3f}, max |chain - pairs| {(chain - pairs).abs().max().item():.3e}, "
          f"sign agreement with the clamping tap-GEMM form {agree:.4f}")
    assert torch.equal(chain, pairs)
    assert agree > 0.9


@pytest.mark.parametrize("B,Tm", [(3, 57), (2, 5), (1, 130)])
def test_fused_pairs_match_two_launch_form_and_oracle(B, Tm):
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch.v1()
    gsd = synth.synth_generator_state(varch)
    mel = synth.synth_mel(B, Tm, 80, 77)
    ref = R.generator_forward(gsd, varch, mel)[:, 0, :]
    fused = _engine(varch, gsd, True).vocode(mel.cuda(), stretch=False).cpu()
    plain = _engine(varch, gsd, False).vocode(mel.cuda(), stretch=False).cpu()
    assert fused.shape == plain.shape == ref.shape == (B, Tm * 256)
    sig = rms(ref)
    e_f, e_p, e_fp = rms(fused, ref), rms(plain, ref), rms(fused, plain)
    print(f"B={B} Tm={Tm}: signal rms {sig:.3f}; fused vs oracle {e_f:.3e}, two-launch vs oracle {e_p:.3e}, fused vs two-launch {e_fp:.3e}")
    assert bool(torch.isfinite(fused).all())
    assert e_f <= 2e-4 and e_p <= 2e-4
    # same operands and roundings, but a different fp32 summation order over (tap, channel chunk): an fp16 rounding that
    # flips early is amplified by the 36 layers behind it (random-weight generator) until the two outputs are as far
    # apart as either is from fp32 -- measured 1.2e-4 for EVERY fused width alone (tools/exp_fuse_compare.py), the
    # round-1 narrow kernels included
    assert e_fp <= 2e-4
    # per clip too: a clip must not depend on its batch neighbours
    if B > 1:
        one = _engine(varch, gsd, True).vocode(mel[1:2].cuda().contiguous(), stretch=False).cpu()
        assert torch.equal(one, fused[1:2])


@pytest.mark.parametrize("arch", ["base", "large"])
def test_lingemm_matches_tapgemm_in_the_bf16_encoder(arch):
    """The dedicated bf16 GEMM kernel (lingemm.hip: Linear layers and the strided feature-extractor convolutions of the
    base model as overlapping-row GEMMs) against the generic tap-GEMM on the same operands (SI_ENC_LINGEMM=0): identical
    bf16 operands and fp32 accumulation, different summation order -> the 80-dim head outputs agree to ~5e-3 relative
    (a bf16 activation copy that rounds the other way is a 2^-9 relative step), far inside the bf16 mode's own error."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch = HubertArch.base() if arch == "base" else HubertArch.large()
    varch = VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(3, 24000, 91).cuda()
    outs = {}
    for flag in ("1", "0"):
        os.environ["SI_ENC_LINGEMM"] = flag
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            os.environ.pop("SI_ENC_LINGEMM", None)
        eng.ctx.profile_start(4000)
        outs[flag] = eng.encode(wave).cpu()
        names = {e["name"] for e in eng.ctx.profile_stop()}
        assert any(n.startswith("lingemm_bf16_") for n in names) == (flag == "1"), names     # the kernel under test actually ran
    ref = InpaintingEngine(harch, varch, 50, "cuda:0", "fp32", "fp32").load_state(hsd, gsd, cb).encode(wave).cpu()
    d = rms(outs["1"], outs["0"]) / rms(outs["0"])
    e1, e0 = rms(outs["1"], ref) / rms(ref), rms(outs["0"], ref) / rms(ref)
    print(f"{arch}: lingemm vs tapgemm {d:.3e} relative; vs the fp32 encoder: lingemm {e1:.3e}, tapgemm {e0:.3e}")
    assert bool(torch.isfinite(outs["1"]).all())
    assert d <= 1.2e-2 and e1 <= 2e-2 and e0 <= 2e-2


@pytest.mark.parametrize("arch,B,N", [("base", 3, 24000), ("large", 2, 16000), ("base", 1, 64000)])
def test_gemm256_matches_lingemm_in_the_bf16_encoder(arch, B, N):
    """The 256 x 256-tile GEMM (gemm256.hip: LDS-DMA staging, two wave groups one barrier apart, persistent tile walk) against the
    128-row kernels on the same bf16 operands: SI_ENC_GEMM256=2 sends EVERY shape it covers through it (feature-extractor
    convolutions with ragged last tiles, both projections, all four Linears of a layer, fp32 + residual and bf16 outputs, GELU
    epilogues), =0 none.  Both kernels order every output's sum identically (K in steps of 32 through the same MFMA with the same
    operand roles, the same epilogue), so the encoder outputs must be EQUAL -- which is what lets the launcher choose between them
    by batch size.  Repeated runs are bit-identical (a race in the DMA / barrier schedule would show as run-to-run noise), and a
    clip does not depend on its batch neighbours."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch = HubertArch.base() if arch == "base" else HubertArch.large()
    varch = VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(B, N, 93).cuda()
    outs, engs = {}, {}
    for flag in ("2", "0"):
        os.environ["SI_ENC_GEMM256"] = flag
        os.environ["SI_ENC_GEMMCU"] = "0"                                   # (the one-tile-per-CU kernel has its own tests below)
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            os.environ.pop("SI_ENC_GEMM256", None)
            os.environ.pop("SI_ENC_GEMMCU", None)
        eng.ctx.profile_start(4000)
        outs[flag] = eng.encode(wave).cpu()
        names = {e["name"] for e in eng.ctx.profile_stop()}
        assert any(n.startswith("gemm256_bf16") for n in names) == (flag == "2"), names
        engs[flag] = eng
    again = [engs["2"].encode(wave).cpu() for _ in range(3)]
    assert all(torch.equal(a, outs["2"]) for a in again)
    if B > 1:
        assert torch.equal(engs["2"].encode(wave[1:2].contiguous()).cpu(), outs["2"][1:2])
    ref = InpaintingEngine(harch, varch, 50, "cuda:0", "fp32", "fp32").load_state(hsd, gsd, cb).encode(wave).cpu()
    e2 = rms(outs["2"], ref) / rms(ref)
    print(f"{arch} B={B} N={N}: gemm256 == 128-row kernels: {torch.equal(outs['2'], outs['0'])}; vs the fp32 encoder {e2:.3e} relative")
    assert bool(torch.isfinite(outs["2"]).all()) and e2 <= 2e-2
    assert torch.equal(outs["2"], outs["0"])


def test_gemm256_persistent_walk_at_the_bench_shape():
    """B = 32 x 4 s (the bench's encoder): the launcher's own rule puts the first four strided convolutions (1600 / 832 / 448 / 256
    tiles: persistent workgroups that request the next tile's first K-tiles under the current tile's last two) and the QKV
    projection on 256 x 256 tiles; the features must equal the all-128-row run bit for bit, twice."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch, varch = HubertArch.base(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(32, 64000, 95).cuda()
    outs = {}
    for flag in ("1", "0"):
        os.environ["SI_ENC_GEMM256"] = flag
        os.environ["SI_ENC_GEMMCU"] = "0"                                   # (gemmcu.hip would take conv4 and the QKV projection first)
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            os.environ.pop("SI_ENC_GEMM256", None)
            os.environ.pop("SI_ENC_GEMMCU", None)
        eng.ctx.profile_start(4000)
        outs[flag] = eng.encode(wave).cpu()
        prof = {e["name"]: e["launches"] for e in eng.ctx.profile_stop()}
        if flag == "1":
            assert prof.get("gemm256_bf16", 0) == 4 + 12, prof                   # conv1-4 + QKV x 12 layers
            assert torch.equal(eng.encode(wave).cpu(), outs["1"])
        else:
            assert "gemm256_bf16" not in prof
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("arch,B,N,flag", [("base", 3, 24000, "10"), ("base", 3, 24000, "11"), ("base", 3, 24000, "12"), ("base", 3, 24000, "13"), ("large", 2, 16000, "14"), ("large", 2, 16000, "15"),
                                           ("large", 2, 16000, "2"), ("base", 1, 64000, "2")])
def test_gemmcu_matches_lingemm_in_the_bf16_encoder(arch, B, N, flag):
    """The one-tile-per-CU GEMM (gemmcu.hip: 16 waves, LDS-DMA ring, tile shape per instantiation) against the 128-row kernels on the
    same bf16 operands.  SI_ENC_GEMMCU=10 + c sends every shape instantiation c covers through it (320 x 256, 256 x 256, 160 x 128, 224 x 128, 128 x 128, 208 x 256: feature-extractor convolutions as per-clip segments with ragged last tiles, both projections, all four Linears of a
    layer, fp32 + residual and bf16 outputs, GELU epilogues), =2 every shape through the instantiation the rule's cost picks, =0
    none.  Same K order through the same MFMA with the same operand roles and epilogue: the encoder outputs must be EQUAL, run to
    run as well (a race in the DMA ring would show as noise), and a clip must not depend on its batch neighbours."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch = HubertArch.base() if arch == "base" else HubertArch.large()
    varch = VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(B, N, 97).cuda()
    outs, engs = {}, {}
    for f in (flag, "0"):
        os.environ["SI_ENC_GEMMCU"] = f
        os.environ["SI_ENC_GEMM256"] = "0"
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            os.environ.pop("SI_ENC_GEMMCU", None)
            os.environ.pop("SI_ENC_GEMM256", None)
        eng.ctx.profile_start(4000)
        outs[f] = eng.encode(wave).cpu()
        names = {e["name"] for e in eng.ctx.profile_stop()}
        assert any(n.startswith("gemmcu_bf16") for n in names) == (f != "0"), names
        engs[f] = eng
    again = [engs[flag].encode(wave).cpu() for _ in range(3)]
    assert all(torch.equal(a, outs[flag]) for a in again)
    if B > 1:
        assert torch.equal(engs[flag].encode(wave[1:2].contiguous()).cpu(), outs[flag][1:2])
    print(f"{arch} B={B} N={N} SI_ENC_GEMMCU={flag}: gemmcu == 128-row kernels: {torch.equal(outs[flag], outs['0'])}")
    assert bool(torch.isfinite(outs[flag]).all())
    assert torch.equal(outs[flag], outs["0"])


@pytest.mark.parametrize("env", [{"SI_ENC_GEMMCU": "0", "SI_ENC_GEMM256": "0"}, {"SI_ENC_GEMMCU": "0", "SI_ENC_GEMM256": "2"}, {"SI_ENC_GEMMCU": "2"}, {}])
def test_layernorm_residual_fusion_changes_no_value(env):
    """Post-LN layers in bf16 mode (SI_ENC_LNFUSE, default 1): a LayerNorm writes its bf16 GEMM operand and (mean, rstd) per row but
    not its fp32 rows; the epilogue of the GEMM that adds them as its residual (out-proj, FFN2 -- in lingemm.hip, gemm256.hip and
    gemmcu.hip) recomputes each element from the row it was normalised from with the LayerNorm kernel's own expression
    (si_ln_apply), and updates the pre-LN sums in place.  The features must be EQUAL to the run in which every LayerNorm writes
    its rows (SI_ENC_LNFUSE=0), with each of the three GEMM kernels forced, and with the launcher's own choice at B = 32."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch, varch = HubertArch.base(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(3 if env else 32, 30000 if env else 64000, 99).cuda()
    outs = {}
    for fuse in ("1", "0"):
        os.environ.update(env)
        os.environ["SI_ENC_LNFUSE"] = fuse
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            for k in list(env) + ["SI_ENC_LNFUSE"]:
                os.environ.pop(k, None)
        outs[fuse] = eng.encode(wave).cpu()
        if fuse == "1":
            assert torch.equal(eng.encode(wave).cpu(), outs["1"])                      # (the in-place update: run to run)
            if env:
                assert torch.equal(eng.encode(wave[1:2].contiguous()).cpu(), outs["1"][1:2])
    assert bool(torch.isfinite(outs["1"]).all())
    assert torch.equal(outs["1"], outs["0"])


def test_ffn_row_padding_changes_no_value():
    """SI_ENC_FFNPAD: the bf16 FFN intermediate is stored with padded rows (default 64 elements: FFN2's operand rows then start
    6272 instead of 6144 bytes apart); a layout choice only -- the encoder output equals the dense layout's bit for bit."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch, varch = HubertArch.base(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(5, 40000, 98).cuda()
    outs = {}
    for pad in ("0", "64", "128"):
        os.environ["SI_ENC_FFNPAD"] = pad
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            os.environ.pop("SI_ENC_FFNPAD", None)
        outs[pad] = eng.encode(wave).cpu()
    assert torch.equal(outs["0"], outs["64"]) and torch.equal(outs["0"], outs["128"])


def test_gemmcu_rule_at_the_bench_shape():
    """B = 32 x 4 s (the bench's encoder, M = 6368 flat rows): the launcher's own rule puts the transformer's Linears and the feature
    projection on one tile per CU (FFN1 on 320 x 256, the N = 768 GEMMs on 160 x 128); the features equal the run without
    the kernel bit for bit, twice."""
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import HubertArch, VocoderArch
    from speech_inpainting_amd.engine import InpaintingEngine
    harch, varch = HubertArch.base(), VocoderArch.tiny()
    hsd, gsd, cb = synth.synth_hubert_state(harch), synth.synth_generator_state(varch), synth.synth_codebook(50)
    wave = synth.synth_wave(32, 64000, 96).cuda()
    outs = {}
    for flag in ("1", "0"):
        os.environ["SI_ENC_GEMMCU"] = flag
        try:
            eng = InpaintingEngine(harch, varch, 50, "cuda:0", "bf16", "fp32").load_state(hsd, gsd, cb)
        finally:
            os.environ.pop("SI_ENC_GEMMCU", None)
        eng.ctx.profile_start(4000)
        outs[flag] = eng.encode(wave).cpu()
        prof = {e["name"]: e["launches"] for e in eng.ctx.profile_stop()}
        cu = sum(v for k, v in prof.items() if k.startswith("gemmcu_bf16"))
        if flag == "1":
            assert cu >= 12 * 3, prof                                           # at least out-proj, FFN1, FFN2 of every layer
            assert torch.equal(eng.encode(wave).cpu(), outs["1"])
        else:
            assert cu == 0, prof
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("scale", [0.125, 1.0, 8.0])
def test_fp16_vocoder_across_activation_scales(scale):
    """The fp16 activation stream at other operating points than the synthetic checkpoint's: `conv_pre` scaled by 1/8 and 8 (the
    stack is positively homogeneous up to the later layers' biases, so every activation in front of the tanh scales with it:
    1/8 puts the stream where the biases dominate, 8 drives the final tanh into saturation on peaks).  fp16 keeps a RELATIVE
    precision, so the waveform error must stay a fixed fraction of the signal: <= 1e-2 of the signal RMS against the fp32
    oracle (measured 1e-3 at scale 1), and inside the 1e-3 absolute gate at scale 1."""
    from oracle import ref_cpu as R
    from speech_inpainting_amd import synth
    from speech_inpainting_amd.arch import VocoderArch
    varch = VocoderArch.v1()
    gsd = synth.synth_generator_state(varch)
    gsd["conv_pre.weight_g"] = gsd["conv_pre.weight_g"] * scale           # w = g v / |v|: scales the folded weight
    gsd["conv_pre.bias"] = gsd["conv_pre.bias"] * scale
    mel = synth.synth_mel(2, 40, 80, 91)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.generator_forward(gsd, varch, mel)[:, 0, :]
    got = _engine(varch, gsd, True).vocode(mel.cuda(), stretch=False).cpu()
    x3 = _engine(varch, gsd, True, voc="bf16x3").vocode(mel.cuda(), stretch=False).cpu()
    sig, err, err3 = rms(ref), rms(got, ref), rms(x3, ref)
    print(f"conv_pre x{scale}: signal rms {sig:.4f} absmax {float(ref.abs().max()):.3f}; fp16 stream error {err:.3e} = {err / sig:.2e} of signal; "
          f"bf16x3 (fp32-equivalent) {err3:.3e}")
    assert bool(torch.isfinite(got).all())
    assert err <= 1e-2 * sig
    assert err3 <= 1e-4 * max(sig, 0.1)
    if scale == 1.0:
        assert err <= 1e-3
