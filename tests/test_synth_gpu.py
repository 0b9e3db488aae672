"""GPU parity of the synthesis stages (HIP path through the C ABI) against oracle/synth.py, same seeded
synthetic weights, same injected randomness.  PARITY UNPINNED w.r.t. the reference (its CosyVoice fork
is not available): these tests pin the HIP kernels to this build's own fp32 CPU restatement.

Stated tolerances (fp16 weights + fp16 MFMA operands with fp32 accumulation vs an all-fp32 oracle), set at ~5x the error
observed on MI355X (every check prints its observed value):
  LM logits / encoders / estimator   max |d| <= 3e-3 * max|ref|   (teacher-forced, every step)
  flow mel                           max |d| <= 5e-3 * max|mel|   after 10 Euler steps x 2 (CFG) estimator passes
  vocoder waveform                   max |d| <= 5e-3 (full scale 0.99), SNR >= 40 dB
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _cfg_and_weights(seed=0):
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny()
    return cfg, make_all(cfg, seed)


# Tolerances: ~5x the error OBSERVED on MI355X (printed by every check), not the loose fp16 worst case
TOL_LOGITS = 3e-3    # LM logits / encoder outputs / estimator, relative to the tensor's scale (observed 4e-4 .. 6e-4)
TOL_MEL = 5e-3       # mel after the full 10-step CFG solve (observed <= 1e-3)
TOL_F0 = 2e-3        # f0 predictor (observed 2e-4)
TOL_WAV = 5e-3       # vocoder waveform, absolute on a 0.99 full scale (observed <= 1e-3)


def _close(got, ref, tol, scale, tag=None):
    err = float((got - ref).abs().max()) / float(scale)
    import inspect
    print(f"[parity] {inspect.stack()[1].function}{'' if tag is None else ' ' + str(tag)}: rel err {err:.2e} (tol {tol:.0e})")
    assert err < tol, (err, tol, tag)


def _snr_db(ref, out):
    ref, out = ref.double(), out.double()
    return float(10 * torch.log10(ref.pow(2).sum() / (ref - out).pow(2).sum().clamp_min(1e-30)))


def test_relpos_encoder_matches_oracle():
    from astts.synth.model import RelPosEncoder
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["flow"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 70, cfg.flow_dim, generator=g)
    lens = torch.tensor([70, 55, 9])
    ref, _ = osyn.relpos_encoder(sd, "encoder", x, lens, cfg.flow_heads, cfg.flow_layers, "swish", ("norm_mha", "norm_ff"),
                                 False, False, cfg.ln_eps, cfg.max_positions)
    enc = RelPosEncoder(sd, "encoder", cfg.flow_heads, cfg.flow_layers, "swish", ("norm_mha", "norm_ff"), False, False,
                        cfg.ln_eps, cfg.max_positions, torch.device(DEV))
    out = enc.forward(x.to(DEV), lens.to(DEV, torch.int32)).cpu()
    for i, L in enumerate(lens.tolist()):
        _close(out[i, :L], ref[i, :L], TOL_LOGITS, float(ref.abs().max()))


def test_lm_teacher_forced_logits_and_sampling():
    from astts.synth.model import AcousticLM
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["llm"]
    g = torch.Generator().manual_seed(1)
    b, tt, tp, steps = 3, 12, 20, 16
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g)
    tlen = torch.full((b,), tt)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    pre_ref = osyn.lm_prefix(sd, cfg, text, tlen, spk, prompt)
    toks_ref, logits_ref = osyn.lm_decode(sd, cfg, pre_ref, steps, u, True, forced)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre = lm.prefix(text.to(DEV), tlen.to(DEV, torch.int32), spk.to(DEV), prompt.to(DEV))
    assert pre.shape == (pre_ref.shape[1], b, cfg.lm_dim)
    _close(pre.cpu().transpose(0, 1), pre_ref, TOL_LOGITS, float(pre_ref.abs().max()))
    scale = float(logits_ref.abs().max())
    for use_engine in (False, True):      # Python-issued fused step, then the C++ decode engine (astts_lm_decode)
        toks, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True, use_engine=use_engine)
        _close(logits.cpu(), logits_ref, TOL_LOGITS, scale)
        assert torch.equal(toks.cpu(), forced.to(torch.int32))
    # free-running: the v1 engine and the Python-issued path take identical kernels in identical order -> identical tokens
    # (the default engine for <= 8 rows is v2, csrc/lm_step.hip: tests/test_lm_step_gpu.py)
    ta = lm.decode(pre, steps, u.to(DEV), True, None, use_engine=False)
    old_env = os.environ.get("ASTTS_LM_ENGINE")
    os.environ["ASTTS_LM_ENGINE"] = "v1"
    try:
        tb = lm.decode(pre, steps, u.to(DEV), True, None, use_engine=True)
    finally:
        if old_env is None:
            os.environ.pop("ASTTS_LM_ENGINE")
        else:
            os.environ["ASTTS_LM_ENGINE"] = old_env
    assert torch.equal(ta, tb)
    # free-running sampling: feed the ORACLE's logits through the HIP sampler step by step (the sampler is
    # exact given identical logits; free-running token equality is not a stable property across precisions)
    from astts import ops
    toks_free, logits_free = osyn.lm_decode(sd, cfg, pre_ref, steps, u, True, None)
    for s in range(steps):
        got = ops.ras_sample(logits_free[:, s].to(DEV), toks_free.to(DEV), s, u[s].to(DEV), cfg.top_k, cfg.top_p, cfg.ras_win,
                             cfg.ras_tau, cfg.speech_vocab, True, eos_policy=cfg.eos_policy).cpu()
        assert got.tolist() == toks_free[:, s].tolist()
    assert int(toks_free.max()) < cfg.speech_vocab          # EOS is never produced in fixed-length decode (either policy)


def test_flow_estimator_and_cfm_match_oracle():
    from astts.synth.model import FlowDecoder
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["flow"]
    g = torch.Generator().manual_seed(2)
    b, tp, ts = 2, 12, 20
    tmp = cfg.mel_frames_for_tokens(tp)
    mel_total = tmp + cfg.mel_frames_for_tokens(ts)
    tokens = torch.randint(0, cfg.speech_vocab, (b, tp + ts), generator=g)
    tlen = torch.full((b,), tp + ts)
    prompt_mel = torch.randn(b, tmp, cfg.mel, generator=g)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    z = torch.randn(b, mel_total, cfg.mel, generator=g)
    fd = FlowDecoder(sd, cfg, torch.device(DEV))
    # one estimator call with ragged lengths (masking semantics)
    x = torch.randn(b, mel_total, cfg.mel, generator=g)
    mu = torch.randn(b, mel_total, cfg.mel, generator=g)
    cond = torch.randn(b, mel_total, cfg.mel, generator=g)
    spk_e = torch.randn(b, cfg.mel, generator=g)
    t = torch.tensor([0.3, 0.8])
    lens = torch.tensor([mel_total, mel_total - 7])
    m = (torch.arange(mel_total)[None, :] < lens[:, None]).float()[..., None]
    ref = osyn.estimator(sd, cfg, x * m, mu * m, spk_e, cond * m, t, lens)
    out = fd.estimator((x * m).to(DEV), (mu * m).to(DEV), spk_e.to(DEV), (cond * m).to(DEV), t.to(DEV), lens.to(DEV, torch.int32)).cpu()
    _close(out, ref, TOL_LOGITS, float(ref.abs().max()))
    # full CFM solve
    ref_mel = osyn.flow_decode(sd, cfg, tokens, tlen, prompt_mel, spk, z, mel_total)
    mel = fd.decode(tokens.to(DEV), tlen.to(DEV, torch.int32), prompt_mel.to(DEV), spk.to(DEV), z.to(DEV), mel_total).cpu()
    assert mel.shape == ref_mel.shape == (b, mel_total - tmp, cfg.mel)
    _close(mel, ref_mel, TOL_MEL, float(ref_mel.abs().max()))


def test_hift_vocoder_matches_oracle():
    from astts.synth.model import HiftVocoder
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["hift"]
    g = torch.Generator().manual_seed(3)
    b, tm = 2, 24
    mel = torch.randn(b, tm, cfg.mel, generator=g)
    nh = cfg.nb_harmonics + 1
    phase0 = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(b, tm * cfg.upsample_total, nh, generator=g)
    voc = HiftVocoder(sd, cfg, torch.device(DEV))
    f0_ref = osyn.hift_f0(sd, cfg, mel)
    f0 = voc.f0(mel.to(DEV)).cpu()
    _close(f0, f0_ref, TOL_F0, float(f0_ref.abs().max()))
    # decode from the SAME source signal (the f0 -> phase map amplifies tiny f0 differences over 6k samples)
    src_ref = osyn.hift_source(sd, cfg, f0_ref, phase0, noise)
    src = voc.source(f0_ref.to(DEV), phase0.to(DEV), noise.to(DEV)).cpu()
    assert float((src - src_ref).abs().max()) < 1e-4
    wav_ref = osyn.hift_decode(sd, cfg, mel, src_ref)
    wav = voc.decode(mel.to(DEV), src_ref.to(DEV)).cpu()
    assert wav.shape == wav_ref.shape == (b, tm * cfg.upsample_total)
    _close(wav, wav_ref, TOL_WAV, 1.0)
    snr = _snr_db(wav_ref, wav)
    print(f"[parity] waveform SNR {snr:.1f} dB")
    assert snr > 40.0
    assert float(wav.abs().max()) <= cfg.audio_limit + 1e-6


@pytest.mark.parametrize("sample_rate", [22050, 24000])
def test_engine_end_to_end_shapes_and_stage_parity(sample_rate):
    """22 050 Hz is what the reference scripts save (tts_with_rag.py:197); 24 000 Hz is the rate `north_star`'s target names
    (`value_24khz` in bench.py): the length regulator then makes 93.75 mel frames per second of tokens instead of 86.13 and the NSF
    source integrates f0 at the other rate -- flow and vocoder held to the oracle at both."""
    import dataclasses

    from astts.synth.model import SynthEngine
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    cfg = dataclasses.replace(cfg, sample_rate=sample_rate)
    eng = SynthEngine(W, cfg, DEV)
    g = torch.Generator().manual_seed(4)
    b, tt, tp, ts = 2, 10, 16, 20
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g)
    tlen = torch.full((b,), tt)
    spk_s, spk_t = torch.randn(b, cfg.spk_dim, generator=g), torch.randn(b, cfg.spk_dim, generator=g)
    style_tok = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    timbre_tok = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    tmp = cfg.mel_frames_for_tokens(tp)
    tm = cfg.mel_frames_for_tokens(ts)
    timbre_mel = torch.randn(b, tmp, cfg.mel, generator=g)
    u = torch.rand(ts, b, 2, generator=g)
    z = torch.randn(b, tmp + tm, cfg.mel, generator=g)
    nh = cfg.nb_harmonics + 1
    phase0 = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(b, tm * cfg.upsample_total, nh, generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, ts), generator=g)
    d = lambda t, dt=None: t.to(DEV) if dt is None else t.to(DEV, dt)
    toks, mel, wav = eng.tts(d(text), d(tlen, torch.int32), d(spk_s), d(style_tok), ts, d(u), d(timbre_tok), d(timbre_mel),
                             d(spk_t), d(z), d(phase0), d(noise), forced_tokens=d(forced))
    assert toks.shape == (b, ts) and mel.shape == (b, tm, cfg.mel) and wav.shape == (b, tm * cfg.upsample_total)
    # oracle chain with the same forced tokens
    all_tok = torch.cat([timbre_tok, forced], dim=1)
    mel_ref = osyn.flow_decode(W["flow"], cfg, all_tok, torch.full((b,), tp + ts), timbre_mel, spk_t, z, tmp + tm)
    _close(mel.cpu(), mel_ref, TOL_MEL, float(mel_ref.abs().max()))
    assert torch.isfinite(wav).all() and float(wav.abs().max()) <= cfg.audio_limit + 1e-6
    assert tm == int(ts / cfg.token_rate * sample_rate / cfg.hop)
    # vocoder at this rate: the oracle's mel and source through both decoders (the f0 -> phase map amplifies f0 differences)
    f0_ref = osyn.hift_f0(W["hift"], cfg, mel_ref)
    src_ref = osyn.hift_source(W["hift"], cfg, f0_ref, phase0, noise)
    src = eng.hift.source(d(f0_ref), d(phase0), d(noise)).cpu()
    assert float((src - src_ref).abs().max()) < 1e-4
    wav_ref = osyn.hift_decode(W["hift"], cfg, mel_ref, src_ref)
    _close(eng.hift.decode(d(mel_ref), d(src_ref)).cpu(), wav_ref, TOL_WAV, 1.0, tag=sample_rate)


def test_pipelined_batches_are_bit_identical_to_sequential():
    """Software pipeline over independent batches (2 decode chains + render stream in flight): every batch must equal
    the sequential result bit for bit, for every batch, repeatedly (guards cross-stream hand-over and buffer reuse)."""
    from astts.synth.model import PipelinedSynth, SynthEngine

    cfg, W = _cfg_and_weights()
    eng = SynthEngine(W, cfg, DEV)
    g = torch.Generator().manual_seed(7)
    b, tt, tp, ts = 4, 10, 16, 40
    d = lambda t, dt=None: t.to(DEV) if dt is None else t.to(DEV, dt)
    tmp, tm = cfg.mel_frames_for_tokens(tp), cfg.mel_frames_for_tokens(ts)
    nh = cfg.nb_harmonics + 1
    batches = []
    for _ in range(3):     # three different batches cycled through the pipeline
        ph = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
        ph[:, 0] = 0
        batches.append((d(torch.randint(0, cfg.text_vocab, (b, tt), generator=g)), d(torch.full((b,), tt), torch.int32),
                        d(torch.randn(b, cfg.spk_dim, generator=g)), d(torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)), ts,
                        d(torch.rand(ts, b, 2, generator=g)), d(torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)),
                        d(torch.randn(b, tmp, cfg.mel, generator=g)), d(torch.randn(b, cfg.spk_dim, generator=g)),
                        d(torch.randn(b, tmp + tm, cfg.mel, generator=g)), d(ph),
                        d(torch.randn(b, tm * cfg.upsample_total, nh, generator=g))))
    refs = [eng.tts(*a) for a in batches]
    torch.cuda.synchronize()
    for depth in (1, 2, 3):
        pipe = PipelinedSynth(eng, lm_depth=depth)
        outs = []
        n = 18
        for i in range(n):
            r = pipe.submit(*batches[i % 3])
            if r is not None:
                outs.append(r)
        outs += pipe.drain()
        torch.cuda.synchronize()
        assert len(outs) == n
        for i, o in enumerate(outs):
            ref = refs[i % 3]
            assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]), (depth, i)


def test_large_batch_is_decoded_in_groups_of_32():
    """B > 32 (BASELINE config 3 shape class): rows are independent, so decoding 40 rows in groups must equal
    decoding each group alone, and the forced-token logits must still match the oracle on a sample of rows."""
    from astts.synth.model import AcousticLM
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["llm"]
    g = torch.Generator().manual_seed(11)
    b, tt, tp, steps = 40, 9, 14, 6
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g)
    tlen = torch.full((b,), tt)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre = lm.prefix(text.to(DEV), tlen.to(DEV, torch.int32), spk.to(DEV), prompt.to(DEV))
    toks, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True)
    assert toks.shape == (b, steps) and torch.equal(toks.cpu(), forced.to(torch.int32))
    rows = [0, 31, 32, 39]
    pre_ref = osyn.lm_prefix(sd, cfg, text[rows], tlen[rows], spk[rows], prompt[rows])
    _, logits_ref = osyn.lm_decode(sd, cfg, pre_ref, steps, u[:, rows], True, forced[rows])
    _close(logits.cpu()[rows], logits_ref, TOL_LOGITS, float(logits_ref.abs().max()))
    free = lm.decode(pre, steps, u.to(DEV), True, None)
    alone = lm.decode(pre[:, 32:].contiguous(), steps, u[:, 32:].contiguous().to(DEV), True, None)
    assert torch.equal(free[32:], alone)


def test_ragged_lm_batch_equals_one_at_a_time():
    """Rows with different text / prompt lengths in ONE left-padded batch must reproduce the oracle run one utterance
    at a time (the only way the reference runs): teacher-forced logits per row, both decode paths."""
    from astts.synth.model import AcousticLM
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["llm"]
    g = torch.Generator().manual_seed(21)
    shapes = [(5, 9), (17, 30), (11, 3), (1, 22)]        # (text tokens, prompt tokens) per row
    steps = 7
    texts = [torch.randint(0, cfg.text_vocab, (tt,), generator=g) for tt, _ in shapes]
    prompts = [torch.randint(0, cfg.speech_vocab, (tp,), generator=g) for _, tp in shapes]
    b = len(shapes)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre, ks = lm.prefix_ragged(texts, spk, prompts)
    assert ks.tolist() == [max(3 + tt + tp for tt, tp in shapes) - (3 + tt + tp) for tt, tp in shapes]
    for use_engine in (False, True):
        toks, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True, use_engine=use_engine, key_start=ks)
        for i in range(b):
            pre_ref = osyn.lm_prefix(sd, cfg, texts[i][None], torch.tensor([shapes[i][0]]), spk[i:i + 1], prompts[i][None])
            _, lref = osyn.lm_decode(sd, cfg, pre_ref, steps, u[:, i:i + 1], True, forced[i:i + 1])
            _close(logits[i].cpu(), lref[0], TOL_LOGITS, float(lref.abs().max()), tag=(use_engine, i))


def test_ragged_flow_batch_equals_one_at_a_time():
    """Flow-matching decode of a ragged batch (different token counts, prompt lengths, output lengths) == each utterance
    decoded alone by the oracle."""
    from astts.synth.model import FlowDecoder
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    sd = W["flow"]
    g = torch.Generator().manual_seed(31)
    shapes = [(12, 20), (5, 33), (21, 8)]                 # (prompt tokens, generated tokens)
    toks, pmels, zs = [], [], []
    for tp, tg in shapes:
        tmp, tm = cfg.mel_frames_for_tokens(tp), cfg.mel_frames_for_tokens(tg)
        toks.append(torch.randint(0, cfg.speech_vocab, (tp + tg,), generator=g))
        pmels.append(torch.randn(tmp, cfg.mel, generator=g))
        zs.append(torch.randn(tmp + tm, cfg.mel, generator=g))
    spk = torch.randn(len(shapes), cfg.spk_dim, generator=g)
    fd = FlowDecoder(sd, cfg, torch.device(DEV))
    mels = fd.decode_ragged(toks, pmels, spk, zs)
    for i, (tp, tg) in enumerate(shapes):
        ref = osyn.flow_decode(sd, cfg, toks[i][None], torch.tensor([tp + tg]), pmels[i][None], spk[i:i + 1], zs[i][None], zs[i].shape[0])[0]
        assert mels[i].shape == ref.shape
        _close(mels[i].cpu(), ref, TOL_MEL, float(ref.abs().max()), tag=i)


@pytest.mark.parametrize("wide", [False, True])
def test_ragged_vocoder_batch_equals_one_at_a_time(wide):
    """HiftVocoder.forward_ragged: utterances of different lengths in ONE vocoder pass (length masks in the implicit-GEMM convolutions,
    the LDS-staged Snake convolutions, the STFT's reflection and the iSTFT's overlap-add) == every utterance vocoded alone, BIT FOR
    BIT, at the toy widths (gemm_tile path everywhere) and at the production widths (conv_lds resblocks; rows shorter and longer than
    one 128-frame tile, so tiles behind a short row's end are skipped beside full ones).  The reference vocodes one utterance at a time
    (tts_with_rag.py:172-197); the one-at-a-time form is held to the oracle by the tests above."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import HiftVocoder
    from astts.synth.weights import make_hift_weights

    cfg = SynthConfig() if wide else SynthConfig.tiny()
    voc = HiftVocoder(make_hift_weights(cfg, 2), cfg, torch.device(DEV))
    g = torch.Generator().manual_seed(77)
    nh, up = cfg.nb_harmonics + 1, cfg.upsample_total
    # rows of MORE than 32 mel frames (every real utterance: 25 tokens = 43 frames is the shortest the drivers ever render) go through
    # the same kernels alone and in a batch: bit-identical.  A row of <= 32 frames run ALONE takes the decode-sized GEMV kernel for its
    # plain (1-tap) products (m <= 32: K split over 8 waves, another summation order), so there the comparison is to rounding -- and the
    # f0 -> phase map turns one ulp of f0 into ~1e-4 of waveform.
    for lens, exact in (([43, 35, 61, 50, 33] if wide else [39, 60, 33, 47], True), ([23, 3, 41, 16, 1] if wide else [9, 30, 2, 17], False)):
        mels = [torch.randn(t, cfg.mel, generator=g).to(DEV) for t in lens]
        ph = []
        for _ in lens:
            p = (torch.rand(1, nh, generator=g) * 2 - 1) * math.pi
            p[:, 0] = 0
            ph.append(p.to(DEV))
        nz = [torch.randn(1, t * up, nh, generator=g).to(DEV) for t in lens]
        got = voc.forward_ragged(mels, ph, nz)
        worst = 0.0
        for j, t in enumerate(lens):
            alone = voc.forward(mels[j][None], ph[j], nz[j])
            assert got[j].shape == alone.shape == (1, t * up)
            assert bool(torch.isfinite(got[j]).all())
            d = float((got[j] - alone).abs().max())
            worst = max(worst, d)
            if exact:
                assert torch.equal(got[j], alone), (j, t, d)
            else:
                assert d < 1e-3, (j, t, d)
        print(f"[parity] ragged vocoder batch vs one at a time ({'wide' if wide else 'tiny'}, lens {lens}): max |d| {worst:.2e}")
    # f0 / source / decode separately with lens == the same calls without lens on a batch of equal lengths (lens is a no-op there)
    same = torch.stack([mels[0], mels[0].flip(0)])
    ll = torch.tensor([lens[0], lens[0]], dtype=torch.int32, device=DEV)
    assert torch.equal(voc.f0(same, ll), voc.f0(same))


@pytest.mark.parametrize("b,t,ragged,wide", [(2, 57, False, False), (3, 64, True, False), (1, 33, True, False), (2, 70, True, True),
                                             (2, 64, False, True)])
def test_flow_solver_engine_is_bit_identical_to_operator_path(b, t, ragged, wide):
    """astts_flow_solve (C++ host loop) issues the same kernels as the operator-by-operator Python solve:
    the solved mel must be identical bit for bit (odd and even T exercise the stride-2 level and its transposed
    convolution; ragged rows exercise every length mask)."""
    from astts.synth.model import FlowDecoder

    cfg, W = _cfg_and_weights()
    if wide:        # 256 estimator channels: the projections with the fused residual + LayerNorm epilogue (astts_op_gemm_ln)
        import dataclasses

        from astts.synth.weights import make_flow_weights

        cfg = dataclasses.replace(cfg, est_channels=(256, 256), est_heads=4, est_mid_blocks=1, est_tfm_per_block=3)
        W = {"flow": make_flow_weights(cfg, 3)}
    fd = FlowDecoder(W["flow"], cfg, torch.device(DEV))
    g = torch.Generator().manual_seed(b * 100 + t)
    z = torch.randn(b, t, cfg.mel, generator=g).to(DEV)
    mu = torch.randn(b, t, cfg.mel, generator=g).to(DEV)
    cond = torch.randn(b, t, cfg.mel, generator=g).to(DEV)
    spk = torch.randn(b, cfg.mel, generator=g).to(DEV)
    lens = None
    if ragged:
        lens = torch.tensor([t, max(t // 2 - 1, 2), t - 3][:b], dtype=torch.int32, device=DEV)
        keep = (torch.arange(t, device=DEV)[None, :] < lens[:, None])[..., None]
        z, mu, cond = z * keep, mu * keep, cond * keep
    ref = fd.solve_ops(z.clone(), mu, spk, cond, lens)
    out = fd.solve(z.clone(), mu, spk, cond, lens)
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref)
    if ragged:
        assert float((out * ~keep).abs().max()) == 0.0


def test_flow_solver_with_more_euler_steps_than_one_time_path_chunk():
    """astts_flow_solve evaluates the time path (embedding -> MLP -> per-ResNet projection) for 32 Euler steps at a time; a solve
    with more steps walks it chunk by chunk (40 = 32 + 8) and must still equal the operator-by-operator solve -- which projects
    all 40 steps' rows in one GEMM, so the comparison allows the rounding of another row tile (fp16 intermediates over 40 Euler
    steps: 1e-4 observed, bar 1e-3 = a fifth of the flow stage's oracle tolerance)."""
    import dataclasses

    from astts.synth.model import FlowDecoder

    cfg, W = _cfg_and_weights()
    cfg = dataclasses.replace(cfg, cfm_steps=40)
    fd = FlowDecoder(W["flow"], cfg, torch.device(DEV))
    g = torch.Generator().manual_seed(77)
    b, t = 2, 45
    z = torch.randn(b, t, cfg.mel, generator=g).to(DEV)
    mu = torch.randn(b, t, cfg.mel, generator=g).to(DEV)
    cond = torch.randn(b, t, cfg.mel, generator=g).to(DEV)
    spk = torch.randn(b, cfg.mel, generator=g).to(DEV)
    ref = fd.solve_ops(z.clone(), mu, spk, cond, None)
    out = fd.solve(z.clone(), mu, spk, cond, None)
    assert torch.isfinite(out).all()
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    print(f"40-step solve, engine vs operator path: rel diff {err:.2e}")
    assert err < 1e-3


def test_no_kernel_writes_outside_its_output_tensor():
    """scripts/oob_check.py guard-bands every tensor the operator wrappers allocate (4 KiB of pattern on both sides)
    and runs LM prefix + decode, the flow decoder (both host paths, fixed and ragged) and the vocoder."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TINY="1", TS="12")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "oob_check.py")], capture_output=True, text=True,
                       env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "guarded tensors" in ln]
    assert len(lines) >= 7 and "OOB WRITE" not in r.stdout, r.stdout[-3000:]
    assert all(ln.rstrip().endswith(" 0 with out-of-bounds writes") for ln in lines), r.stdout[-3000:]
    assert "engine == ops: True" in r.stdout


def test_fullsize_pipeline_and_cobatching_are_bit_identical_to_sequential():
    """Full-size model (the shapes where the decode GEMMs take the split-K path and workgroups of concurrent chains land
    on different XCDs from launch to launch): batches pushed through the stream pipeline -- separate decode chains, and
    two batches co-batched into one 16-row chain -- give the same tokens, mel and waveform, bit for bit, as one batch run
    alone.  (A stale-L2 read in the split-K reduction showed up exactly here and nowhere in single-stream runs.)"""
    from astts.synth.config import SynthConfig
    from astts.synth.model import PipelinedSynth, SynthEngine
    from astts.synth.weights import make_all

    cfg = SynthConfig()
    eng = SynthEngine(make_all(cfg, 0), cfg, DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    B, Tt, Tp, Ts = 8, 24, 60, 40
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
    args = (text, tlen, spk_s, style_tok, Ts, u, timbre_tok, timbre_mel, spk_t, z, phase0, noise)
    ref = eng.tts(*args)
    torch.cuda.synchronize()
    # front_prefill: the LM prefill enqueued on the caller's stream (default) or on the decode chain with the steps
    for depth, cob, fp in ((2, 1, True), (2, 1, False), (2, 2, True), (1, 3, False)):
        pipe = PipelinedSynth(eng, lm_depth=depth, lm_priority=0, render_priority=0, cobatch=cob, front_prefill=fp)
        assert pipe.front_prefill is fp
        outs = []
        with torch.cuda.stream(pipe.front_stream):
            for _ in range(9):
                r = pipe.submit(*args)
                if r is not None:
                    outs.append(r)
            outs += pipe.drain()
        torch.cuda.synchronize()
        assert len(outs) == 9
        for o in outs:
            assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]), (depth, cob, fp)


def test_fullsize_lm_logits_match_oracle():
    """CosyVoice-300M shapes (d = 1024, 14 layers, FFN 4096): the decode step's kernels at the sizes the benchmark runs
    -- MFMA relative-position prefill, fused LayerNorm + QKV into the fp16 KV cache, split-K FFN-out, output head --
    against the fp32 oracle, teacher-forced over a few steps.  Tolerance as stated for the tiny model (TOL_LOGITS of the logit
    scale: fp16 weights / operands / KV cache vs all-fp32)."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_lm_weights
    from oracle import synth as osyn

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    g = torch.Generator().manual_seed(11)
    b, tt, tp, steps = 2, 9, 14, 4
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g)
    tlen = torch.full((b,), tt)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    pre_ref = osyn.lm_prefix(sd, cfg, text, tlen, spk, prompt)
    _, logits_ref = osyn.lm_decode(sd, cfg, pre_ref, steps, u, True, forced)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre = lm.prefix(text.to(DEV), tlen.to(DEV, torch.int32), spk.to(DEV), prompt.to(DEV))
    _close(pre.cpu().transpose(0, 1), pre_ref, TOL_LOGITS, float(pre_ref.abs().max()))
    toks, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True)
    assert torch.equal(toks.cpu(), forced.to(torch.int32))
    _close(logits.cpu(), logits_ref, TOL_LOGITS, float(logits_ref.abs().max()))


def test_fullsize_flow_estimator_matches_oracle():
    """One evaluation of the full-size U-Net estimator (256 channels, 12 mid blocks x 4 transformer blocks: the ring GEMMs,
    flash attention, fused GroupNorm at the benchmark's widths) against the fp32 oracle on a short ragged batch."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import FlowDecoder
    from astts.synth.weights import make_flow_weights
    from oracle import synth as osyn

    cfg = SynthConfig()
    sd = make_flow_weights(cfg, 1)
    g = torch.Generator().manual_seed(12)
    b, t = 2, 70
    x = torch.randn(b, t, cfg.mel, generator=g)
    mu = torch.randn(b, t, cfg.mel, generator=g)
    cond = torch.randn(b, t, cfg.mel, generator=g)
    spk_e = torch.randn(b, cfg.mel, generator=g)
    tt = torch.tensor([0.25, 0.7])
    lens = torch.tensor([t, t - 9])
    m = (torch.arange(t)[None, :] < lens[:, None]).float()[..., None]
    ref = osyn.estimator(sd, cfg, x * m, mu * m, spk_e, cond * m, tt, lens)
    fd = FlowDecoder(sd, cfg, torch.device(DEV))
    out = fd.estimator((x * m).to(DEV), (mu * m).to(DEV), spk_e.to(DEV), (cond * m).to(DEV), tt.to(DEV), lens.to(DEV, torch.int32)).cpu()
    _close(out, ref, TOL_LOGITS, float(ref.abs().max()))


def test_fullsize_hift_vocoder_matches_oracle():
    """HiFT at its real widths (512 -> 256 -> 128 channels, x256 upsampling) on a short mel: f0, NSF source and the
    conv-transpose / Snake-resblock stack + iSTFT against the fp32 oracle (same tolerances as the tiny model)."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import HiftVocoder
    from astts.synth.weights import make_hift_weights
    from oracle import synth as osyn

    cfg = SynthConfig()
    sd = make_hift_weights(cfg, 2)
    g = torch.Generator().manual_seed(13)
    b, tm = 1, 16
    mel = torch.randn(b, tm, cfg.mel, generator=g)
    nh = cfg.nb_harmonics + 1
    phase0 = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(b, tm * cfg.upsample_total, nh, generator=g)
    voc = HiftVocoder(sd, cfg, torch.device(DEV))
    f0_ref = osyn.hift_f0(sd, cfg, mel)
    f0 = voc.f0(mel.to(DEV)).cpu()
    _close(f0, f0_ref, TOL_F0, float(f0_ref.abs().max()))
    src_ref = osyn.hift_source(sd, cfg, f0_ref, phase0, noise)
    src = voc.source(f0_ref.to(DEV), phase0.to(DEV), noise.to(DEV)).cpu()
    assert float((src - src_ref).abs().max()) < 1e-4
    wav_ref = osyn.hift_decode(sd, cfg, mel, src_ref)
    wav = voc.decode(mel.to(DEV), src_ref.to(DEV)).cpu()
    assert wav.shape == wav_ref.shape == (b, tm * cfg.upsample_total)
    _close(wav, wav_ref, TOL_WAV, 1.0)
    snr = _snr_db(wav_ref, wav)
    print(f"[parity] waveform SNR {snr:.1f} dB")
    assert snr > 40.0


def test_stream_render_stage_parity_and_chunk_shapes():
    """stream=True (astts.synth.stream): the oracle runs the chunked schedule (3 chunks: 230 tokens = hops at 0 and 100 + 30 left
    over) and records every stage call; each HIP stage is held to the oracle on the oracle's own inputs at every call (the
    f0 -> phase map amplifies 1e-4 f0 differences over 50 k samples, so stages are compared on identical inputs, as in the
    one-shot tests); then the HIP engine runs the same schedule end to end: same chunk shapes, finite, clamped."""
    from astts.synth.model import FlowDecoder, HiftVocoder
    from astts.synth.stream import StreamConsts, stream_render
    from oracle import synth as osyn

    cfg, W = _cfg_and_weights()
    g = torch.Generator().manual_seed(21)
    tp, n_tok = 16, 230
    tmp = cfg.mel_frames_for_tokens(tp)
    ptok = torch.randint(0, cfg.speech_vocab, (1, tp), generator=g)
    pmel = torch.randn(1, tmp, cfg.mel, generator=g)
    spk = torch.randn(1, cfg.spk_dim, generator=g)
    toks = torch.randint(0, cfg.speech_vocab, (n_tok,), generator=g)
    nh = cfg.nb_harmonics + 1
    consts = StreamConsts.for_config(cfg)
    draws = {"z": [], "src": []}
    rec = {"flow": [], "f0": [], "src": [], "voc": []}

    def o_flow(tok):
        n = cfg.mel_frames_for_tokens(int(tok.numel()))
        z = torch.randn(1, tmp + n, cfg.mel, generator=g)
        draws["z"].append(z)
        all_tok = torch.cat([ptok, tok.view(1, -1)], 1)
        mel = osyn.flow_decode(W["flow"], cfg, all_tok, torch.tensor([all_tok.shape[1]]), pmel, spk, z, tmp + n)
        rec["flow"].append((tok.clone(), mel))
        return mel

    def o_f0(mel):
        f = osyn.hift_f0(W["hift"], cfg, mel)
        rec["f0"].append((mel.clone(), f))
        return f

    def o_src(f):
        ph = (torch.rand(1, nh, generator=g) * 2 - 1) * math.pi
        ph[:, 0] = 0
        nz = torch.randn(1, f.shape[1] * cfg.upsample_total, nh, generator=g)
        draws["src"].append((ph, nz))
        s = osyn.hift_source(W["hift"], cfg, f, ph, nz)
        rec["src"].append((f.clone(), s))
        return s

    def o_voc(mel, s):
        w = osyn.hift_decode(W["hift"], cfg, mel, s)
        rec["voc"].append((mel.clone(), s.clone(), w))
        return w

    ref_chunks = list(stream_render(toks, consts, o_flow, o_f0, o_src, o_voc))
    assert len(ref_chunks) == 3 and [int(t.numel()) for t, _ in rec["flow"]] == [120, 120, 30]

    dev = torch.device(DEV)
    fd, voc = FlowDecoder(W["flow"], cfg, dev), HiftVocoder(W["hift"], cfg, dev)
    # stage parity on the oracle's inputs, call by call
    for i, (tok, mel_ref) in enumerate(rec["flow"]):
        all_tok = torch.cat([ptok, tok.view(1, -1)], 1).to(dev, torch.int32)
        mel = fd.decode(all_tok, torch.tensor([all_tok.shape[1]], dtype=torch.int32, device=dev), pmel.to(dev), spk.to(dev),
                        draws["z"][i].to(dev), tmp + mel_ref.shape[1]).cpu()
        _close(mel, mel_ref, TOL_MEL, float(mel_ref.abs().max()), f"flow chunk {i}")
    for i, (mel_in, f_ref) in enumerate(rec["f0"]):
        _close(voc.f0(mel_in.to(dev)).cpu(), f_ref, TOL_F0, float(f_ref.abs().max()), f"f0 chunk {i}")
    for i, (f_in, s_ref) in enumerate(rec["src"]):
        ph, nz = draws["src"][i]
        assert float((voc.source(f_in.to(dev), ph.to(dev), nz.to(dev)).cpu() - s_ref).abs().max()) < 1e-4
    for i, (mel_in, s_in, w_ref) in enumerate(rec["voc"]):
        w = voc.decode(mel_in.to(dev), s_in.to(dev)).cpu()
        _close(w, w_ref, TOL_WAV, 1.0, f"vocoder chunk {i}")
        assert _snr_db(w_ref, w) > 40.0
    # the product schedule end to end on the HIP stages (same draws)
    it = {"z": iter(draws["z"]), "src": iter(draws["src"])}

    def h_flow(tok):
        z = next(it["z"])
        all_tok = torch.cat([ptok, tok.view(1, -1)], 1).to(dev, torch.int32)
        return fd.decode(all_tok, torch.tensor([all_tok.shape[1]], dtype=torch.int32, device=dev), pmel.to(dev), spk.to(dev), z.to(dev),
                         z.shape[1])

    def h_src(f):
        ph, nz = next(it["src"])
        return voc.source(f, ph.to(dev), nz.to(dev))

    chunks = [c.cpu() for c in stream_render(toks, consts, h_flow, voc.f0, h_src, voc.decode)]
    assert [tuple(c.shape) for c in chunks] == [tuple(c.shape) for c in ref_chunks]
    for c in chunks:
        assert bool(torch.isfinite(c).all()) and float(c.abs().max()) <= cfg.audio_limit + 1e-6
    total = sum(int(c.shape[1]) for c in chunks)
    print(f"[parity] stream: {len(chunks)} chunks, {total} samples for {n_tok} tokens ({n_tok / cfg.token_rate:.2f} s -> {total / cfg.sample_rate:.2f} s)")
    assert abs(total / cfg.sample_rate - n_tok / cfg.token_rate) < 0.1


def test_hift_vocoder_production_widths_match_oracle():
    """The vocoder at the PRODUCTION channel widths (hift_base 512: 256- and 128-channel resblocks -> the LDS-staged Snake
    convolution ops.conv1d_snake, resblock mean and source sum folded into conv epilogues) on a short mel against the oracle, same
    source signal as in test_hift_vocoder_matches_oracle."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import HiftVocoder
    from astts.synth.weights import make_hift_weights
    from oracle import synth as osyn

    cfg = SynthConfig()
    sd = make_hift_weights(cfg, 3)
    g = torch.Generator().manual_seed(33)
    b, tm = 2, 9
    mel = torch.randn(b, tm, cfg.mel, generator=g)
    nh = cfg.nb_harmonics + 1
    phase0 = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(b, tm * cfg.upsample_total, nh, generator=g)
    voc = HiftVocoder(sd, cfg, torch.device(DEV))
    assert all(rb.lds for grp in voc.res for rb in grp) and all(rb.lds for rb in voc.sres)
    f0_ref = osyn.hift_f0(sd, cfg, mel)
    src_ref = osyn.hift_source(sd, cfg, f0_ref, phase0, noise)
    wav_ref = osyn.hift_decode(sd, cfg, mel, src_ref)
    wav = voc.decode(mel.to(DEV), src_ref.to(DEV)).cpu()
    assert wav.shape == wav_ref.shape == (b, tm * cfg.upsample_total)
    _close(wav, wav_ref, TOL_WAV, 1.0)
    snr = _snr_db(wav_ref, wav)
    print(f"[parity] waveform SNR {snr:.1f} dB")
    assert snr > 40.0 and float(wav.abs().max()) <= cfg.audio_limit + 1e-6


def test_stream_pipe_classes_and_autotuned_pipeline():
    """ops.stream_pipe_classes groups streams by command-processor pipe with a launch-chain probe (queues that share a pipe run
    chains ~2.4x slower each); PipelinedSynth.autotune builds its pipeline on one stream per pipe.  Properties: every candidate
    lands in exactly one class, two chains on streams of DIFFERENT classes do not slow each other, and the autotuned pipeline
    returns the sequential results bit for bit."""
    import threading
    import time

    from astts import _lib, ops
    from astts.synth.model import PipelinedSynth, SynthEngine

    dev = torch.device(DEV)
    classes = ops.stream_pipe_classes(candidates=8, device=dev)
    flat = [s for c in classes for s in c]
    assert len(flat) == 8 and len({s.cuda_stream for s in flat}) == 8 and 1 <= len(classes) <= 8
    lib = _lib.load()

    def chains(group, count=300):
        torch.cuda.synchronize()

        def one(st):
            _lib.check(lib.astts_stream_chain(count, 3, 64, int(st.cuda_stream)))
            st.synchronize()
        th = [threading.Thread(target=one, args=(s,)) for s in group]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return (time.perf_counter() - t0) / count

    if len(classes) >= 2:
        a, b = classes[0][0], classes[1][0]
        chains([a]); chains([b])
        alone = max(min(chains([a]), chains([a])), min(chains([b]), chains([b])))
        both = min(chains([a, b]), chains([a, b]))
        print(f"[pipes] {len(classes)} pipes; chain alone {alone * 1e6:.1f} us / launch, two chains on two pipes {both * 1e6:.1f} us / launch")
        assert both < 1.6 * alone
    cfg, W = _cfg_and_weights()
    eng = SynthEngine(W, cfg, DEV)
    g = torch.Generator().manual_seed(17)
    b, tt, tp, ts = 2, 10, 16, 24
    d = lambda t, dt=None: t.to(DEV) if dt is None else t.to(DEV, dt)
    tmp, tm = cfg.mel_frames_for_tokens(tp), cfg.mel_frames_for_tokens(ts)
    nh = cfg.nb_harmonics + 1
    ph = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
    ph[:, 0] = 0
    args = (d(torch.randint(0, cfg.text_vocab, (b, tt), generator=g)), d(torch.full((b,), tt), torch.int32),
            d(torch.randn(b, cfg.spk_dim, generator=g)), d(torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)), ts,
            d(torch.rand(ts, b, 2, generator=g)), d(torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)),
            d(torch.randn(b, tmp, cfg.mel, generator=g)), d(torch.randn(b, cfg.spk_dim, generator=g)),
            d(torch.randn(b, tmp + tm, cfg.mel, generator=g)), d(ph), d(torch.randn(b, tm * cfg.upsample_total, nh, generator=g)))
    ref = eng.tts(*args)
    torch.cuda.synchronize()
    pipe = PipelinedSynth.autotune(eng, args, depths=(2, 3), trials=1, steps=2)
    assert pipe.tuned_ms_per_batch > 0 and pipe.front_stream is not None
    streams = [pipe.s_render, *pipe.s_lm]
    assert len({s.cuda_stream for s in streams}) == len(streams)
    with torch.cuda.stream(pipe.front_stream):
        outs = []
        for _ in range(5):
            r = pipe.submit(*args)
            if r is not None:
                outs.append(r)
        outs += pipe.drain()
    torch.cuda.synchronize()
    assert len(outs) == 5
    for o in outs:
        assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])
