"""Parity at the shapes, tiles and batch sizes the BENCHMARK runs: the tile of a ring GEMM is picked
by row count, so a test on 140 rows does not touch the kernels that serve 5 504 / 11 008 rows.

  * ring GEMM (fp16 activations) on the flow decoder's projection shapes, through the automatic rule AND forced through
    every ring tile (astts_op_gemm_set_ring_mode), against an fp32 reference of the same fp16-rounded operands;
  * one full-shape estimator pass at BASELINE config-2 geometry (B=1 -> cond + uncond, T = 688 frames) vs the oracle;
  * BASELINE config 3 at full size (64 rows x (Tt=64, Ts=250)): rows 32..63 of the 64-row batch equal the same rows run as
    their own batch (tokens bit for bit given the same prefix values), waveform finite and clamped;
  * BASELINE config 5's second bank: 100k x 768, Q=256, against the oracle on a sample + size-independent properties.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (rows, k, n, activation, fp16 output, residual): the estimator's transformer projections at 5504 rows (T=344 x 16) and
# 11008 rows (T=688 x 16), + one shape with 2048 128x128 tiles (the long-form tile rule)
RING_SHAPES = [
    (5504, 256, 1536, "none", True, False),      # fused q|k|v (3 x 512)
    (5504, 256, 1024, "gelu", True, False),      # FFN-in
    (5504, 512, 256, "none", False, True),       # attention out-proj + residual
    (5504, 1024, 256, "none", False, True),      # FFN-out + residual
    (11008, 256, 1536, "none", True, False),
    (11008, 1024, 256, "none", False, True),
    (32768, 256, 1024, "gelu", True, False),     # 256 x 8 = 2048 tiles of 128x128: ring mode 1 by the rule
    (5520, 256, 512, "none", False, False),      # ragged last row tile
    (5000, 1024, 4097, "none", False, True),     # 20 x 17 = 340 tiles of 256x256 (forced modes 4 / 5; 128x128 by the rule: 0.66 of two rounds); ragged in both directions
    (3800, 2048, 3000, "none", True, False),     # 15 x 12 = 180 tiles of 256x256 = 0.70 of a round at K = 2048: the eight-phase tile by the rule; ragged
]


@pytest.mark.parametrize("mode", [-1, 1, 2, 3, 4, 5, 0])
@pytest.mark.parametrize("shape", RING_SHAPES, ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}-{s[3]}")
def test_ring_gemm_bench_shapes_every_tile(shape, mode):
    from astts import ops

    m, k, n, act, o16, res = shape
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).half()
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g) if res else None
    pw = ops.PackedWeight(w, b)
    ops.set_gemm_ring_mode(mode)
    try:
        y = ops.linear(x.to(DEV), pw, act=act, residual=None if r is None else r.to(DEV),
                       out_dtype=torch.float16 if o16 else torch.float32)
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_ring_mode(-1)
    ref = F.linear(x.double(), w.half().double(), b.double())        # same fp16-rounded operands, exact accumulation
    ref = {"none": lambda t: t, "gelu": F.gelu}[act](ref)
    if res:
        ref = ref + r.double()
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    # fp32 accumulation of <= 1024 products + A&S erf (2e-7): 1e-5; an fp16 output adds its own rounding (2^-11 relative)
    tol = 1.5e-3 if o16 else 2e-5
    print(f"ring mode {mode} {shape}: rel err {err:.2e} (tol {tol:.1e})")
    assert err < tol
    assert y.dtype == (torch.float16 if o16 else torch.float32)


@pytest.mark.parametrize("shape", [(256, 64, 256), (300, 192, 260), (5000, 1024, 4097), (8192, 2048, 2048), (256, 6144, 20000), (40000, 256, 1536)],
                         ids=lambda s: "x".join(map(str, s)))
def test_eight_phase_ring_is_bit_identical_to_the_one_barrier_ring(shape):
    """gemm_ring8 (the launcher's 256 x 256 tile from round 6; forced: ring mode 5) multiplies the same tile with the same accumulation
    order per accumulator as gemm_ring<4, 2, 2, 4, 8> (forced: mode 4), so their outputs are equal bit for bit; its LDS-DMA data is ordered
    only by counted vmcnt + barriers, so the comparison repeats on fresh inputs (a missed wait shows as a flicker).  One, two, three K
    tiles (prologue / tail paths of the six-deep request pipeline), ragged edges, a single row panel, shallow K with many tiles."""
    from astts import ops

    m, k, n = shape
    g = torch.Generator().manual_seed(m * 7 + n)
    w, b = torch.randn(n, k, generator=g) / 8, torch.randn(n, generator=g) * 0.1
    pw = ops.PackedWeight(w, b)
    try:
        for rep in range(6):
            x = torch.randn(m, k, generator=g).half()
            xd = x.to(DEV)
            ops.set_gemm_ring_mode(4)
            y4 = ops.gemm(xd, pw, out=torch.empty((m, n), dtype=torch.float32, device=DEV))
            ops.set_gemm_ring_mode(5)
            y5 = ops.gemm(xd, pw, out=torch.empty((m, n), dtype=torch.float32, device=DEV))
            torch.cuda.synchronize()
            assert torch.equal(y4, y5), f"rep {rep}: {int((y4 != y5).sum())} elements differ"
            if rep == 0:       # (and both are right: exact accumulation of the same fp16-rounded operands)
                rows = torch.randperm(m, generator=g)[:64]
                ref = F.linear(x[rows].double(), w.half().double(), b.double())
                assert float((y5[rows.to(DEV)].double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-5
    finally:
        ops.set_gemm_ring_mode(-1)


def test_fullshape_estimator_pass_config2_geometry():
    """B=1 utterance of the benchmark: T = 258 prompt + 430 generated = 688 frames, cond + uncond as one 2-row batch --
    the row counts (1376 at full rate, 688 after the down block) and every kernel variant of one estimator evaluation at
    the benchmark's sequence length, against the fp32 oracle."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import FlowDecoder
    from astts.synth.weights import make_flow_weights
    from oracle import synth as osyn

    cfg = SynthConfig()
    sd = make_flow_weights(cfg, 1)
    g = torch.Generator().manual_seed(42)
    t = 258 + 430
    x = torch.randn(1, t, cfg.mel, generator=g)
    mu = torch.randn(1, t, cfg.mel, generator=g)
    cond = torch.zeros(1, t, cfg.mel)
    cond[:, :258] = torch.randn(1, 258, cfg.mel, generator=g)
    spk_e = torch.randn(1, cfg.mel, generator=g)
    # the CFG pair exactly as the solver forms it: row 0 conditioned, row 1 with mu / spk / cond zeroed
    x2 = torch.cat([x, x], 0)
    mu2 = torch.cat([mu, torch.zeros_like(mu)], 0)
    spk2 = torch.cat([spk_e, torch.zeros_like(spk_e)], 0)
    cond2 = torch.cat([cond, torch.zeros_like(cond)], 0)
    tt = torch.tensor([0.3, 0.3])
    lens = torch.tensor([t, t])
    torch.set_num_threads(min(torch.get_num_threads(), 64))
    ref = osyn.estimator(sd, cfg, x2, mu2, spk2, cond2, tt, lens)
    fd = FlowDecoder(sd, cfg, torch.device(DEV))
    for full in (True, False):       # the fixed-length fast path (no length masks launched) and the masked path
        out = fd.estimator(x2.to(DEV), mu2.to(DEV), spk2.to(DEV), cond2.to(DEV), tt.to(DEV), lens.to(DEV, torch.int32), full=full).cpu()
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        print(f"full-shape estimator (T={t}, full={full}): rel err vs oracle {err:.2e}")
        assert err < 5e-3


def test_config3_longform_batch_rows_are_batch_independent():
    """BASELINE config 3 at full size: 64 rows x (Tt=64 text tokens, Ts=250 speech tokens) in ONE engine call.  The LM
    decodes 64 rows as two 32-row groups on two streams, flow + vocoder take the 64-row batch: rows 32..63 must equal
    the same rows run as their own 32-row batch -- tokens bit for bit, mel and waveform to the rounding noise of a
    different GEMM tile -- and the waveform is finite and clamped."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import SynthEngine
    from astts.synth.weights import make_all

    cfg = SynthConfig()
    eng = SynthEngine(make_all(cfg, 0), cfg, DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    B, Tt, Tp, Ts = 64, 64, 150, 250
    text = torch.randint(0, cfg.text_vocab, (B, Tt), device=DEV, generator=g)
    tlen = torch.full((B,), Tt, dtype=torch.int32, device=DEV)
    spk_s = torch.randn(B, cfg.spk_dim, device=DEV, generator=g)
    spk_t = torch.randn(B, cfg.spk_dim, device=DEV, generator=g)
    style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=DEV, generator=g)
    timbre_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=DEV, generator=g)
    tmp, tm = cfg.mel_frames_for_tokens(Tp), cfg.mel_frames_for_tokens(Ts)
    timbre_mel = torch.randn(B, tmp, cfg.mel, device=DEV, generator=g)
    u = torch.rand(Ts, B, 2, device=DEV, generator=g)
    z = torch.randn(B, tmp + tm, cfg.mel, device=DEV, generator=g)
    nh = cfg.nb_harmonics + 1
    phase0 = (torch.rand(B, nh, device=DEV, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(B, tm * cfg.upsample_total, nh, device=DEV, generator=g)
    pre = eng.lm.prefix(text, tlen, spk_s, style_tok)            # text encoder + prompt embedding, all 64 rows
    toks = eng.lm.decode(pre, Ts, u, ignore_eos=True)            # two 32-row groups, concurrent on two streams
    mel, wav = eng.tts_render(toks, timbre_tok, timbre_mel, spk_t, z, phase0, noise)
    torch.cuda.synchronize()
    assert toks.shape == (B, Ts) and wav.shape == (B, tm * cfg.upsample_total)
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= cfg.audio_limit + 1e-6
    assert int(toks.max()) < cfg.speech_vocab and int(toks.min()) >= 0
    sl = slice(32, 64)
    # the same rows as their own batch.  The prefix VALUES are shared: the text encoder picks its GEMM tile by row count, and a
    # different tile sums the same products in a different order (1 ulp), which free-running sampling amplifies over 250 steps
    # -- true of any batched implementation; the claim here is about the decode groups, the flow stage and the vocoder.
    toks2 = eng.lm.decode(pre[:, sl].contiguous(), Ts, u[:, sl].contiguous(), ignore_eos=True)
    assert torch.equal(toks2, toks[sl])
    pre32 = eng.lm.prefix(text[sl], tlen[sl], spk_s[sl], style_tok[sl])
    dp = float((pre32 - pre[:, sl]).abs().max()) / float(pre.abs().max())
    print(f"config 3: prefix of rows 32..63 computed in a 32-row vs the 64-row batch: rel diff {dp:.1e}")
    assert dp < 1e-6
    mel2, wav2 = eng.tts_render(toks2, timbre_tok[sl], timbre_mel[sl], spk_t[sl], z[sl], phase0[sl], noise[sl])
    torch.cuda.synchronize()
    dm = float((mel2 - mel[sl]).abs().max()) / float(mel.abs().max())
    dw = float((wav2 - wav[sl]).abs().max())
    print(f"config 3: rows 32..63 alone vs inside the 64-row batch: mel rel diff {dm:.2e}, wav abs diff {dw:.2e}")
    # a 64-row batch takes other GEMM tiles than a 32-row one (tile by row count): same products, other fp32 summation order,
    # and the fp16 intermediates of 10 Euler steps x 2 x 56 transformer blocks round differently now and then.  Bar: an
    # order of magnitude inside the oracle tolerance of the flow stage (5e-3 of the mel scale).
    # The waveform: the vocoder turns a mel difference of 4e-4 into isolated sample differences of a few 1e-3 (full scale 0.99);
    # the bar is the one the synthesis parity test puts on the vocoder against the oracle (SNR > 40 dB) plus a peak bound.
    snr = 10.0 * math.log10(float((wav[sl].double() ** 2).sum()) / max(float(((wav2 - wav[sl]).double() ** 2).sum()), 1e-30))
    print(f"config 3: waveform SNR of the two evaluations {snr:.1f} dB")
    assert dm < 1e-3 and dw < 1e-2 and snr > 40.0


def test_config5_bank_100k_x_768():
    """BASELINE config 5's Moka-dimension bank: 100k x 768 (fp16-exact Gaussian rows), Q=256, k=3."""
    from astts.knn import StyleBank
    from oracle import knn as oknn

    n, d, nq, k = 100_000, 768, 256, 3
    g = torch.Generator(device=DEV).manual_seed(1234)
    bank_t = torch.randn((n, d), generator=g, device=DEV, dtype=torch.float32).to(torch.float16)
    sb = StyleBank(bank_t)
    rows = torch.randint(0, n, (nq,), generator=g, device=DEV)
    q = bank_t[rows].to(torch.float32)
    idx, sc = sb.search_device(q, k)
    torch.cuda.synchronize()
    assert torch.equal(idx[:, 0], rows) and torch.allclose(sc[:, 0], torch.ones(nq, device=DEV), atol=1e-6)
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    noisy = q + 0.7 * torch.randn(q.shape, generator=g, device=DEV)
    i_fast, s_fast = (t.clone() for t in sb.search_device(noisy, k))
    nfb = sb.last_fallbacks()
    i_ex, s_ex = sb.search_device(noisy, k, force_exact=True)
    assert torch.equal(i_fast, i_ex) and torch.equal(s_fast, s_ex)
    print(f"100k x 768, Q=256: {nfb} queries took the exact fallback")
    bank_np = bank_t.cpu().numpy()
    ei, es = oknn.knn_search(bank_np, noisy[:32].cpu().numpy(), k)          # 32 queries x 100k x 768 in fp64: a few seconds
    assert np.array_equal(i_fast[:32].cpu().numpy(), ei)
    assert np.allclose(s_fast[:32].cpu().numpy(), es, atol=1e-6, rtol=0)
    # small query group (register-streaming scan instead of the GEMM scan) returns the same answers
    i8, s8 = sb.search_device(noisy[:8].contiguous(), k)
    assert torch.equal(i8, i_fast[:8]) and torch.equal(s8, s_fast[:8])
