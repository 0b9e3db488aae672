"""Decode engine "v3" (csrc/lm_fused.hip: two launches per layer -- attention block per (head, row pair), feed-forward block per 64
hidden features -- with the residual stream in 64-bit fixed point, partial sums met by integer atomics) against the fp32 oracle and
against engine v2 under teacher forcing.  Stated tolerance vs the oracle: 3e-3 of the logit scale, as for v2 (fp16 weights / operands /
KV cache vs all-fp32); v3 vs v2 share every operand rounding except where fp32 partial sums are rounded: 1e-3.  Integer accumulation is
associative, so v3 must also be bit-reproducible and independent of the batch width -- checked with torch.equal."""
import pytest
import torch

from test_lm_step_gpu import DEV, _engine, _forced_case, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b", [1, 2, 3, 8, 12, 17, 32])
def test_tiny_v3_logits_match_oracle_and_v2(b):
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny()
    sd = make_all(cfg, 0)["llm"]
    steps = 9
    lm, pre, u, forced, ref = _setup(cfg, sd, b, 7, 11, steps, 100 + b)
    scale = float(ref.abs().max())
    out = {}
    for eng in ("v2", "v3"):
        with _engine(eng):
            toks, logits = lm.decode(pre, steps, u, True, forced.to(DEV), return_logits=True)
        assert torch.equal(toks.cpu(), forced.to(torch.int32))
        out[eng] = logits.cpu()
        err = float((out[eng] - ref).abs().max()) / scale
        print(f"tiny b={b} {eng}: logits rel err vs oracle {err:.2e}")
        assert err < 3e-3
    d = float((out["v3"] - out["v2"]).abs().max()) / scale
    print(f"tiny b={b}: v3 vs v2 {d:.2e}")
    assert d < 1e-3
    assert not torch.equal(out["v3"][:, 1:], out["v2"][:, 1:]), "v3 was not selected (the logits equal v2's bit for bit)"


def test_tiny_v3_ragged_rows_match_oracle_one_at_a_time():
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_all
    from oracle import synth as osyn

    cfg = SynthConfig.tiny()
    sd = make_all(cfg, 0)["llm"]
    g = torch.Generator().manual_seed(5)
    shapes = [(5, 9), (17, 30), (11, 3), (1, 22), (8, 8)]
    steps = 8
    texts = [torch.randint(0, cfg.text_vocab, (tt,), generator=g) for tt, _ in shapes]
    prompts = [torch.randint(0, cfg.speech_vocab, (tp,), generator=g) for _, tp in shapes]
    b = len(shapes)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre, ks = lm.prefix_ragged(texts, spk, prompts)
    with _engine("v3"):
        _, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True, key_start=ks)
    for i in range(b):
        pre_ref = osyn.lm_prefix(sd, cfg, texts[i][None], torch.tensor([shapes[i][0]]), spk[i:i + 1], prompts[i][None])
        _, lref = osyn.lm_decode(sd, cfg, pre_ref, steps, u[:, i:i + 1], True, forced[i:i + 1])
        err = float((logits[i].cpu() - lref[0]).abs().max()) / float(lref.abs().max())
        print(f"ragged row {i}: {err:.2e}")
        assert err < 3e-3, i


def test_fullsize_v3_bench_geometry_every_step_vs_oracle():
    """BASELINE config 2's decode (B=8, Tt=32, Tp=150, 250 tokens, CosyVoice-300M widths): logits of all 250 steps."""
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_lm_weights

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    err, s0 = _forced_case(cfg, sd, 8, 32, 150, 250, 31, "v3")
    assert s0 == 185
    print(f"bench geometry v3: keys {s0}..{s0 + 249}; logits rel err vs oracle max {float(err.max()):.2e} (step {int(err.argmax())}), "
          f"first {float(err[0]):.2e}, last {float(err[-1]):.2e}")
    assert float(err.max()) < 3e-3


@pytest.mark.parametrize("b,tp", [(3, 1100), (8, 1300), (32, 1690)])
def test_tiny_v3_long_context_vs_oracle(b, tp):
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny().with_(max_positions=2048)
    sd = make_all(cfg, 0)["llm"]
    steps = 6
    err, s0 = _forced_case(cfg, sd, b, 7, tp, steps, 500 + b, "v3")
    print(f"tiny long context v3 b={b}: keys {s0}..{s0 + steps - 1}; logits rel err vs oracle {float(err.max()):.2e}")
    assert float(err.max()) < 3e-3


@pytest.mark.parametrize("rows", [8, 13, 32])
def test_fullsize_v3_is_reproducible_and_rows_do_not_depend_on_the_batch_width(rows):
    """Integer atomics: the order the 16 heads / 64 hidden slices arrive in cannot change a bit.  The same call twice gives the same
    logits; rows taken out of a wide batch give the logits they had inside it (17+ rows use two MFMA row tiles in the feed-forward
    block, odd row counts leave half a row pair empty in the attention block)."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_lm_weights

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    g = torch.Generator().manual_seed(90 + rows)
    tt, tp, steps = 20, 130, 10
    text = torch.randint(0, cfg.text_vocab, (rows, tt), generator=g).to(DEV)
    tlen = torch.full((rows,), tt, dtype=torch.int32, device=DEV)
    spk = torch.randn(rows, cfg.spk_dim, generator=g).to(DEV)
    prompt = torch.randint(0, cfg.speech_vocab, (rows, tp), generator=g).to(DEV)
    forced = torch.randint(0, cfg.speech_vocab, (rows, steps), generator=g).to(DEV)
    u = torch.rand(steps, rows, 2, generator=g).to(DEV)
    pre = lm.prefix(text, tlen, spk, prompt)
    with _engine("v3"):
        _, wide = lm.decode(pre, steps, u, True, forced, return_logits=True)
        _, again = lm.decode(pre, steps, u, True, forced, return_logits=True)
        assert torch.equal(wide, again)
        for sl in (slice(0, 5), slice(rows - 3, rows), slice(1, 2)):
            _, narrow = lm.decode(pre[:, sl].contiguous(), steps, u[:, sl].contiguous(), True, forced[sl].contiguous(), return_logits=True)
            assert torch.equal(narrow, wide[sl]), (rows, sl)
        t_w = lm.decode(pre, steps, u, True, None)
        t_n = lm.decode(pre[:, :4].contiguous(), steps, u[:, :4].contiguous(), True, None)
        assert torch.equal(t_n, t_w[:4])
