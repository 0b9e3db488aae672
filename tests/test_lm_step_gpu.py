"""The decode-step kernels of csrc/lm_step.hip (engine "v2": 8-column diagonal-MFMA GEMVs, one-round-trip attention with
key split merged by its consumer, embedding LayerNorm inside layer 0's QKV kernel, LayerNorm scale/shift folded into
the weights) against the fp32 oracle AND against the operator chain ("v1", the engine of round 1) under teacher
forcing.  Stated tolerance vs the oracle: 3e-3 of the logit scale (~5x the observed 5e-4) (fp16 weights / operands / KV cache vs all-fp32);
observed values are printed.  v2 vs v1 share every operand rounding except the summation order: 1e-3."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _engine:
    def __init__(self, which):
        self.which = which

    def __enter__(self):
        self.old = os.environ.get("ASTTS_LM_ENGINE")
        os.environ["ASTTS_LM_ENGINE"] = self.which

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("ASTTS_LM_ENGINE", None)
        else:
            os.environ["ASTTS_LM_ENGINE"] = self.old


def _setup(cfg, sd, b, tt, tp, steps, seed):
    from astts.synth.model import AcousticLM
    from oracle import synth as osyn

    g = torch.Generator().manual_seed(seed)
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
    return lm, pre, u.to(DEV), forced, logits_ref


@pytest.mark.parametrize("b", [1, 3, 8, 12, 20])
def test_tiny_v2_logits_match_oracle_and_v1(b):
    """b <= 8: diagonal 8-column form; 12: 16-column form; 20: two row tiles.  (v2 is forced for b > 8; the default
    engine choice keeps v1 there.)"""
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny()
    sd = make_all(cfg, 0)["llm"]
    steps = 9
    lm, pre, u, forced, ref = _setup(cfg, sd, b, 7, 11, steps, 100 + b)
    scale = float(ref.abs().max())
    out = {}
    for eng in ("v1", "v2"):
        with _engine(eng):
            toks, logits = lm.decode(pre, steps, u, True, forced.to(DEV), return_logits=True)
        assert torch.equal(toks.cpu(), forced.to(torch.int32))
        out[eng] = logits.cpu()
        err = float((out[eng] - ref).abs().max()) / scale
        print(f"tiny b={b} {eng}: logits rel err vs oracle {err:.2e}")
        assert err < 3e-3
    d12 = float((out["v1"] - out["v2"]).abs().max()) / scale
    print(f"tiny b={b}: v2 vs v1 {d12:.2e}")
    assert d12 < 1e-3
    # free running (sampling on the device): the two engines see logits that differ by ~1e-3 of their scale, so tokens agree
    # except at near-ties of the sampler's inverse CDF; require the first steps to agree and every token to be valid
    with _engine("v1"):
        t1 = lm.decode(pre, steps, u, True, None)
    with _engine("v2"):
        t2 = lm.decode(pre, steps, u, True, None)
    assert int(t2.max()) < cfg.speech_vocab and int(t2.min()) >= 0
    assert torch.equal(t1[:, 0], t2[:, 0])          # step 0 samples the (shared) prefill logits
    agree = float((t1 == t2).float().mean())
    print(f"tiny b={b}: free-running token agreement v1/v2 {agree:.3f}")


def test_tiny_v2_ragged_rows_match_oracle_one_at_a_time():
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
    with _engine("v2"):
        _, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True, key_start=ks)
    for i in range(b):
        pre_ref = osyn.lm_prefix(sd, cfg, texts[i][None], torch.tensor([shapes[i][0]]), spk[i:i + 1], prompts[i][None])
        _, lref = osyn.lm_decode(sd, cfg, pre_ref, steps, u[:, i:i + 1], True, forced[i:i + 1])
        err = float((logits[i].cpu() - lref[0]).abs().max()) / float(lref.abs().max())
        print(f"ragged row {i}: {err:.2e}")
        assert err < 3e-3, i


def test_fullsize_v2_logits_match_oracle_and_v1_with_long_context():
    """CosyVoice-300M widths, 8 rows (the benchmark batch), a prefix long enough that the key-split attention has both
    halves populated and more than one 64-key pass per wave (>= 130 keys)."""
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_lm_weights

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    steps = 4
    lm, pre, u, forced, ref = _setup(cfg, sd, 8, 24, 120, steps, 77)
    scale = float(ref.abs().max())
    out = {}
    for eng in ("v1", "v2"):
        with _engine(eng):
            toks, logits = lm.decode(pre, steps, u, True, forced.to(DEV), return_logits=True)
        out[eng] = logits.cpu()
        err = float((out[eng] - ref).abs().max()) / scale
        print(f"full size b=8 {eng}: logits rel err vs oracle {err:.2e}")
        assert err < 3e-3
    d12 = float((out["v1"] - out["v2"]).abs().max()) / scale
    print(f"full size: v2 vs v1 {d12:.2e}")
    assert d12 < 1e-3


def test_fullsize_v2_is_deterministic_and_row_independent():
    """Same call twice -> the same bits (fixed summation orders everywhere); and a row's tokens do not depend on which rows
    sit next to it (the property the pipeline's co-batching relies on)."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_lm_weights

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    g = torch.Generator().manual_seed(9)
    b, tt, tp, steps = 8, 16, 40, 12
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g).to(DEV)
    tlen = torch.full((b,), tt, dtype=torch.int32, device=DEV)
    spk = torch.randn(b, cfg.spk_dim, generator=g).to(DEV)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g).to(DEV)
    u = torch.rand(steps, b, 2, generator=g).to(DEV)
    pre = lm.prefix(text, tlen, spk, prompt)
    with _engine("v2"):
        t1 = lm.decode(pre, steps, u, True, None)
        t2 = lm.decode(pre, steps, u, True, None)
        assert torch.equal(t1, t2)
        sub = [1, 4, 6]
        t3 = lm.decode(pre[:, sub].contiguous(), steps, u[:, sub].contiguous(), True, None)
        assert torch.equal(t3, t1[sub])


# ---------------------------------------------------------------------------------------------------------------------
# Long-context regimes.  The oracle's teacher-forced logits of EVERY step come from causal passes over
# prefix (+) emb(forced[:-1]) (query chunks of 256 against the caches of the earlier chunks: the oracle's relative-position
# tensor is tq x tk x d), not from a 250-step loop: seconds on the host.
def _oracle_forced_logits(sd, cfg, text, tlen, spk, prompt, forced, chunk=256):
    from oracle import synth as osyn

    pre = osyn.lm_prefix(sd, cfg, text, tlen, spk, prompt)
    s0 = pre.shape[1]
    emb = sd["speech_embedding.weight"][forced[:, :-1].long()]
    x = torch.cat([pre, emb], 1)
    caches, outs = None, []
    with torch.no_grad():
        for t0 in range(0, x.shape[1], chunk):
            lg, caches = osyn.lm_forward(sd, cfg, x[:, t0:t0 + chunk], caches)
            outs.append(lg)
    return torch.cat(outs, 1)[:, s0 - 1:], s0          # [B, steps, V + 1]: step s is predicted from position s0 - 1 + s


def _forced_case(cfg, sd, b, tt, tp, steps, seed, engine):
    from astts.synth.model import AcousticLM

    g = torch.Generator().manual_seed(seed)
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g)
    tlen = torch.full((b,), tt)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    ref, s0 = _oracle_forced_logits(sd, cfg, text, tlen, spk, prompt, forced)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre = lm.prefix(text.to(DEV), tlen.to(DEV, torch.int32), spk.to(DEV), prompt.to(DEV))
    with _engine(engine):
        toks, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True)
    assert torch.equal(toks.cpu(), forced.to(torch.int32))
    scale = float(ref.abs().max())
    err = (logits.cpu() - ref).abs().amax(dim=(0, 2)) / scale          # per step
    return err, s0


def test_fullsize_v2_bench_geometry_every_step_vs_oracle():
    """The benchmark's decode (BASELINE config 2: B=8, Tt=32, Tp=150, 250 tokens, CosyVoice-300M widths): 184 -> 433 keys,
    i.e. up to four 64-key passes per key-split half of lm_attn.  Logits of all 250 steps against the oracle."""
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_lm_weights

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    err, s0 = _forced_case(cfg, sd, 8, 32, 150, 250, 31, "v2")
    assert s0 == 185
    print(f"bench geometry v2: keys {s0}..{s0 + 249}; logits rel err vs oracle max {float(err.max()):.2e} (step {int(err.argmax())}), "
          f"first {float(err[0]):.2e}, last {float(err[-1]):.2e}")
    assert float(err.max()) < 3e-3


@pytest.mark.parametrize("engine,b,tp", [("v2", 3, 1100), ("v2", 8, 1300), ("v1", 32, 1690), ("v2", 32, 1690), ("v2", 16, 1100)])
def test_tiny_long_context_vs_oracle(engine, b, tp):
    """> 1 024 keys: each key-split half of lm_attn walks more than one 256-key chunk (csrc/lm_step.hip chunk loop); v1's
    attn_relpos_decode at the ~1 700 keys x 32 rows of the long-form probe (BASELINE config 3); the same for the wide v2 forms."""
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny().with_(max_positions=2048)
    sd = make_all(cfg, 0)["llm"]
    steps = 6
    err, s0 = _forced_case(cfg, sd, b, 7, tp, steps, 500 + b, engine)
    print(f"tiny long context {engine} b={b}: keys {s0}..{s0 + steps - 1}; logits rel err vs oracle {float(err.max()):.2e}")
    assert float(err.max()) < 3e-3


@pytest.mark.parametrize("rows", [12, 16, 20, 32])
def test_fullsize_v2_rows_do_not_depend_on_the_batch_width(rows):
    """The decode step's arithmetic for a row is the same in 8-, 16- and 32-row launches (csrc/lm_step.hip: the projection
    form is chosen by shape, the halved 8-column forms sum in the order of the diagonal form, attention is per (row, head)):
    teacher-forced logits of rows 0..7 taken from a wide batch equal those of the 8-row batch BIT FOR BIT, at CosyVoice-300M
    widths, with a context long enough for both key halves of lm_attn."""
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
    pre = lm.prefix(text, tlen, spk, prompt)                      # shared prefix VALUES (the prefill picks GEMM tiles by row count)
    with _engine("v2"):
        _, wide = lm.decode(pre, steps, u, True, forced, return_logits=True)
        for sl in (slice(0, 8), slice(rows - 8, rows), slice(3, 7)):
            _, narrow = lm.decode(pre[:, sl].contiguous(), steps, u[:, sl].contiguous(), True, forced[sl].contiguous(), return_logits=True)
            assert torch.equal(narrow, wide[sl]), (rows, sl)
        # free running as well: tokens of the rows agree
        t_w = lm.decode(pre, steps, u, True, None)
        t_n = lm.decode(pre[:, :8].contiguous(), steps, u[:, :8].contiguous(), True, None)
        assert torch.equal(t_n, t_w[:8])


def test_embedding_table_path_equals_the_per_step_projection(monkeypatch):
    """The decode step gathers its input row from the table speech_emb W^T + b formed at load; ASTTS_LM_EMBED_TABLE=0 leaves the
    table out and the engine projects the sampled token's embedding every step (the branch no caller reaches otherwise).  Same fp16
    products, another fp32 summation order (big-tile GEMM vs the step's GEMV): teacher-forced logits agree to rounding, free-running
    tokens agree."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_lm_weights

    cfg = SynthConfig()
    sd = make_lm_weights(cfg, 0)
    g = torch.Generator().manual_seed(77)
    b, tt, tp, steps = 8, 16, 40, 12
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g).to(DEV)
    tlen = torch.full((b,), tt, dtype=torch.int32, device=DEV)
    spk = torch.randn(b, cfg.spk_dim, generator=g).to(DEV)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g).to(DEV)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g).to(DEV)
    u = torch.rand(steps, b, 2, generator=g).to(DEV)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ASTTS_LM_EMBED_TABLE", flag)
        lm = AcousticLM(sd, cfg, torch.device(DEV))
        pre = lm.prefix(text, tlen, spk, prompt)
        with _engine("v2"):
            _, lg = lm.decode(pre, steps, u, True, forced, return_logits=True)
            tk = lm.decode(pre, steps, u, True, None)
        assert (lm.embed_table is None) == (flag == "0")
        outs[flag] = (lg.cpu(), tk.cpu())
    scale = float(outs["1"][0].abs().max())
    err = float((outs["1"][0] - outs["0"][0]).abs().max()) / scale
    print(f"embedding table vs per-step projection: logits rel diff {err:.2e}")
    assert err < 1e-3
    assert torch.equal(outs["1"][1], outs["0"][1])


def test_group_steps_stop_each_32_row_group_at_its_own_length():
    """AcousticLM.decode(group_steps=): a batch wider than 32 rows runs as 32-row chains; group g decodes only group_steps[g] steps
    (ragged batches sorted by length).  Its tokens equal the unrestricted decode's first group_steps[g] columns; the rest is zero."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny()
    lm = AcousticLM(make_all(cfg, 0)["llm"], cfg, torch.device(DEV))
    g = torch.Generator().manual_seed(5)
    b, tt, tp, steps = 40, 6, 9, 12
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g).to(DEV)
    tlen = torch.full((b,), tt, dtype=torch.int32, device=DEV)
    spk = torch.randn(b, cfg.spk_dim, generator=g).to(DEV)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g).to(DEV)
    u = torch.rand(steps, b, 2, generator=g).to(DEV)
    pre = lm.prefix(text, tlen, spk, prompt)
    full = lm.decode(pre, steps, u, True, None).cpu()
    part = lm.decode(pre, steps, u, True, None, group_steps=[steps, 5]).cpu()
    assert torch.equal(part[:32], full[:32])
    assert torch.equal(part[32:, :5], full[32:, :5]) and int(part[32:, 5:].abs().sum()) == 0
    with pytest.raises(ValueError):
        lm.decode(pre, steps, u, True, None, group_steps=[steps])


@pytest.mark.parametrize("size,b", [("tiny", 40), ("tiny", 97), ("full", 72)])
def test_wide_engine_logits_match_oracle(size, b):
    """Batches of 33 .. 256 rows on the engine's WIDE path (csrc/lm_engine.hip decode_wide: one plain GEMM per projection for all rows --
    the ring kernel on the fp16 LayerNorm output / FFN hidden -- K|V straight into the cache row, per-row decode attention): teacher-forced
    logits of every step within the usual 3e-3 of the oracle (rows sampled for the oracle: it runs one row in ~a second at full size),
    free-running tokens valid, ragged rows (key_start) included, and steps issued in two ranges == one range (astts_lm_decode_range)."""
    from astts.synth.config import SynthConfig
    from astts.synth.model import AcousticLM
    from astts.synth.weights import make_all, make_lm_weights
    from oracle import synth as osyn

    cfg = SynthConfig.tiny() if size == "tiny" else SynthConfig()
    sd = make_all(cfg, 0)["llm"] if size == "tiny" else make_lm_weights(cfg, 0)
    steps = 7 if size == "tiny" else 5
    g = torch.Generator().manual_seed(900 + b)
    # ragged rows: text lengths 3 .. 12, prompt lengths 5 .. 40 (full: up to 150)
    tmax, pmax = 12, (40 if size == "tiny" else 150)
    tls = torch.randint(3, tmax + 1, (b,), generator=g).tolist()
    pls = torch.randint(5, pmax + 1, (b,), generator=g).tolist()
    texts = [torch.randint(0, cfg.text_vocab, (n,), generator=g) for n in tls]
    prompts = [torch.randint(0, cfg.speech_vocab, (n,), generator=g) for n in pls]
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g)
    u = torch.rand(steps, b, 2, generator=g)
    lm = AcousticLM(sd, cfg, torch.device(DEV))
    pre, ks = lm.prefix_ragged(texts, spk, prompts)
    toks, logits = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True, key_start=ks, wide=True)
    assert torch.equal(toks.cpu(), forced.to(torch.int32))
    rows = list(range(b)) if size == "tiny" else [0, b // 2, b - 1]
    worst = 0.0
    for i in rows:
        pre_ref = osyn.lm_prefix(sd, cfg, texts[i][None], torch.tensor([tls[i]]), spk[i:i + 1], prompts[i][None])
        _, lref = osyn.lm_decode(sd, cfg, pre_ref, steps, u[:, i:i + 1], True, forced[i:i + 1])
        err = float((logits[i].cpu() - lref[0]).abs().max()) / float(lref.abs().max())
        worst = max(worst, err)
        assert err < 3e-3, (i, err)
    print(f"[parity] wide engine {size} b={b}: logits rel err vs oracle (rows {len(rows)}) {worst:.2e}")
    # the same rows through the 32-row groups (the default for b > 32): logits agree to rounding
    _, l32 = lm.decode(pre, steps, u.to(DEV), True, forced.to(DEV), return_logits=True, key_start=ks)
    d = float((logits - l32).abs().max()) / float(l32.abs().max())
    print(f"[parity] wide engine {size} b={b}: vs the 32-row groups {d:.2e}")
    assert d < 2e-3
    # free running: valid tokens, step 0 (sampled from the shared prefill logits) equal to the grouped path's
    t_w = lm.decode(pre, steps, u.to(DEV), True, None, key_start=ks, wide=True)
    t_g = lm.decode(pre, steps, u.to(DEV), True, None, key_start=ks)
    assert int(t_w.max()) < cfg.speech_vocab and int(t_w.min()) >= 0 and torch.equal(t_w[:, 0], t_g[:, 0])
    # two ranges of steps == one
    st = lm.prefill(pre, steps, ks)
    ctx = lm.decode_begin(st, u.to(DEV), True, None)
    lm.decode_range(ctx, 3)
    lm.decode_range(ctx)
    assert torch.equal(ctx["toks"], t_w)
    # inside the wide engine a row does not depend on the rows beside it (gemm_rows / attn_relpos_rows sum a row's terms in an order of
    # its own): the first 36 rows as a wide batch of their own
    if b >= 72:
        n2 = 36
        _, lsub = lm.decode(pre[:, :n2].contiguous(), steps, u[:, :n2].contiguous().to(DEV), True, forced[:n2].to(DEV), return_logits=True,
                            key_start=ks[:n2].contiguous(), wide=True)
        dsub = float((lsub - logits[:n2]).abs().max())
        print(f"[parity] wide engine {size} b={b}: rows 0..{n2 - 1} as their own wide batch differ by {dsub:.3e}")
        assert dsub == 0.0
