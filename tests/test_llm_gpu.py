"""GPU parity of the query embedder (astts.llm.embedder.LlamaEmbedder, csrc/ops_llm.hip + the GEMM family) against the
golden fixtures produced by transformers' LlamaForCausalLM in fp32 (tests/golden/make_llama_fixtures.py) on the seeded
weights of astts.llm.weights.make_llama_weights.  Tolerance: fp16 weights / MFMA operands vs fp32 -- 5e-3 of the tensor's
scale for hidden states and embeddings (observed values are printed); greedy tokens equal."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def _load(name):
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    fx = np.load(os.path.join(GOLD, f"llama_{name}.npz"))
    cfg = getattr(LlamaShape, name)()
    return fx, cfg, LlamaEmbedder(make_llama_weights(cfg, int(fx["seed"])), cfg, DEV)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("name", ["tiny", "wide"])
def test_embedder_matches_transformers_fixtures(name):
    fx, cfg, emb = _load(name)
    ids, lens = torch.from_numpy(fx["ids"]), fx["lens"]
    row0 = ids[:1, :lens[0]]
    h = emb.hidden(row0).cpu().numpy()[0]
    e_h = _rel(h, fx["hidden_final_row0"])
    print(f"[parity] llama {name}: final hidden rel err {e_h:.2e}")
    assert e_h < 1e-2          # K up to 8192 at the real widths: observed 4e-3
    one = np.stack([emb.embed_ids(ids[i:i + 1, :n]).cpu().numpy()[0] for i, n in enumerate(lens)])     # one text per call, as the reference
    e_1 = _rel(one, fx["embedding"])
    batched = emb.embed_ids(ids, torch.from_numpy(lens)).cpu().numpy()                                   # right-padded batch
    e_b = _rel(batched, fx["embedding"])
    print(f"[parity] llama {name}: mean-pooled embedding rel err {e_1:.2e} (one at a time), {e_b:.2e} (padded batch)")
    assert e_1 < 5e-3 and e_b < 5e-3
    assert one.dtype == np.float32 and one.shape == (len(lens), cfg.hidden)
    lg = emb.logits_last(row0.to(DEV)).cpu().numpy()[0]
    e_l = _rel(lg, fx["logits_last_row0"])
    print(f"[parity] llama {name}: last-token logits rel err {e_l:.2e}")
    assert e_l < 1e-2
    n_new = len(fx["greedy"]) - int(lens[0])
    gen = emb.generate_greedy(row0[0].tolist(), n_new)                  # KV cache, argmax on the device, one synchronisation
    ref = fx["greedy"].tolist()
    # greedy tokens: equal unless the fp32 reference itself has a near-tie at a step (top-2 logit gap below the fp16 error)
    if gen != ref:
        first = next(i for i, (a, b) in enumerate(zip(gen, ref)) if a != b)
        pytest.fail(f"greedy continuation differs from transformers at position {first}: {gen} vs {ref}")
    assert emb.generate_greedy_recompute(row0[0].tolist(), n_new) == ref          # the prompt-per-token form (second implementation)
    # a BATCH of prompts of different lengths (left-padded, per-row RoPE shift + key_start mask): every row equals its own run
    prompts = [ids[i, :int(n)].tolist() for i, n in enumerate(lens)]
    batch = emb.generate_greedy_batch(prompts, n_new)
    assert batch[0] == ref
    for p_, got in zip(prompts, batch):
        assert got == emb.generate_greedy_recompute(p_, n_new), (len(p_), got)
    # the VALU attention (rounds 3-4) and the MFMA attention give the same embedding to fp16 rounding
    emb.mfma_attention = False
    e_v = _rel(emb.embed_ids(ids, torch.from_numpy(lens)).cpu().numpy(), fx["embedding"])
    emb.mfma_attention = True
    print(f"[parity] llama {name}: padded-batch embedding rel err with the VALU attention {e_v:.2e} (MFMA {e_b:.2e})")
    assert e_v < 5e-3


def test_llm_operators_against_definitions():
    """rmsnorm / rope / causal GQA attention / swiglu / mean-pool, each against its fp32 torch definition (oracle/llama.py)."""
    from astts import ops
    from astts.llm.config import LlamaShape
    from oracle import llama as ol

    cfg = LlamaShape.tiny()
    g = torch.Generator().manual_seed(3)
    b, t, heads, kvh, hd = 2, 77, 4, 2, 128
    x = torch.randn(b, t, 512, generator=g) * 3
    w = 1 + 0.1 * torch.randn(512, generator=g)
    y = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5, out_dtype=torch.float32).cpu()
    assert _rel(y, ol.rmsnorm(x, w, 1e-5)) < 1e-5
    qkv = torch.randn(b, t, (heads + 2 * kvh) * hd, generator=g).half()
    cos, sin = ol.rope_tables(cfg, t + 5)
    q = qkv[..., :heads * hd].float().view(b, t, heads, hd).transpose(1, 2)
    k = qkv[..., heads * hd:(heads + kvh) * hd].float().view(b, t, kvh, hd).transpose(1, 2)
    v = qkv[..., (heads + kvh) * hd:].float().view(b, t, kvh, hd).transpose(1, 2)
    qr = q * cos[:t] + ol._rotate_half(q) * sin[:t]
    kr = k * cos[:t] + ol._rotate_half(k) * sin[:t]
    d = qkv.to(DEV).clone()
    ops.rope_llama_(d, cos[:, :hd // 2].contiguous().to(DEV), sin[:, :hd // 2].contiguous().to(DEV), heads + kvh, hd)
    got_q = d[..., :heads * hd].float().cpu().view(b, t, heads, hd).transpose(1, 2)
    got_k = d[..., heads * hd:(heads + kvh) * hd].float().cpu().view(b, t, kvh, hd).transpose(1, 2)
    assert _rel(got_q, qr) < 2e-3 and _rel(got_k, kr) < 2e-3                     # fp16 storage of the rotated values
    assert torch.equal(d[..., (heads + kvh) * hd:].cpu(), qkv[..., (heads + kvh) * hd:])   # v untouched
    lens = torch.tensor([t, 40])
    mask = torch.full((t, t), float("-inf")).triu(1)[None, None].expand(b, 1, t, t).clone()
    mask = mask.masked_fill((torch.arange(t)[None, :] >= lens[:, None])[:, None, None, :], float("-inf"))
    rep = heads // kvh
    s = got_q @ got_k.repeat_interleave(rep, 1).transpose(-1, -2) / hd ** 0.5 + mask
    ref = (torch.softmax(s, -1) @ v.repeat_interleave(rep, 1)).transpose(1, 2).reshape(b, t, heads * hd)
    a = ops.attn_causal_gqa(d[..., :heads * hd], d[..., heads * hd:(heads + kvh) * hd], d[..., (heads + kvh) * hd:], heads, kvh, hd,
                            lens.to(DEV, torch.int32)).float().cpu()
    am = ops.attn_gqa(d[..., :heads * hd], d[..., heads * hd:(heads + kvh) * hd], d[..., (heads + kvh) * hd:], heads, kvh, hd,
                      lens=lens.to(DEV, torch.int32)).float().cpu()                # the same on the matrix cores (attn_gqa_mfma)
    for i in range(b):
        n = int(lens[i])
        assert _rel(a[i, :n], ref[i, :n]) < 2e-3, i
        assert _rel(am[i, :n], ref[i, :n]) < 2e-3, i
        assert float(a[i, n:].abs().max()) == 0.0 if n < t else True              # padded queries produce zeros
        assert float(am[i, n:].abs().max()) == 0.0 if n < t else True
    # generation-path form: time-major cache views, left padding (key_start), a block of new queries at key index pos0 (tq < tk),
    # more than one 128-query block and more than one 64-key tile
    for (tq, tk) in ((1, 201), (3, 77), (150, 333)):
        pos0 = tk - tq
        gq = torch.randn(tq, b, heads * hd, generator=g).half()
        gkv = torch.randn(tk, b, 2 * kvh * hd, generator=g).half()
        ks = torch.tensor([0, min(37, tk - tq)], dtype=torch.int32)
        qh = gq.float().view(tq, b, heads, hd).permute(1, 2, 0, 3)
        kh = gkv[..., :kvh * hd].float().view(tk, b, kvh, hd).permute(1, 2, 0, 3).repeat_interleave(rep, 1)
        vh = gkv[..., kvh * hd:].float().view(tk, b, kvh, hd).permute(1, 2, 0, 3).repeat_interleave(rep, 1)
        kidx, qidx = torch.arange(tk)[None, None, None, :], (pos0 + torch.arange(tq))[None, None, :, None]
        bad = (kidx > qidx) | (kidx < ks[:, None, None, None])
        sc = (qh @ kh.transpose(-1, -2) / hd ** 0.5).masked_fill(bad, float("-inf"))
        want = (torch.softmax(sc, -1) @ vh).permute(2, 0, 1, 3).reshape(tq, b, heads * hd)
        dkv = gkv.to(DEV)
        got = ops.attn_gqa(gq.to(DEV), dkv[..., :kvh * hd], dkv[..., kvh * hd:], heads, kvh, hd, key_start=ks.to(DEV), pos0=pos0,
                           time_major=True).float().cpu()
        valid = (pos0 + torch.arange(tq))[:, None] >= ks[None, :]                  # a pad query (before its row's first token) has no key: zeros
        assert _rel(got[valid], want[valid]) < 2e-3, (tq, tk)
        assert float(got[~valid].abs().max()) == 0.0 if bool((~valid).any()) else True
    # rope with a per-row shift on a time-major buffer == the plain kernel on each row's own (unpadded) sequence
    tmx = torch.randn(t, b, (heads + kvh) * hd, generator=g).half()
    sh = torch.tensor([0, 9], dtype=torch.int32)
    r_tm = ops.rope_llama_ex_(tmx.to(DEV).clone(), cos[:, :hd // 2].contiguous().to(DEV), sin[:, :hd // 2].contiguous().to(DEV), heads + kvh, hd,
                              pos0=0, shift=sh.to(DEV), time_major=True).cpu()
    for i in range(b):
        own = tmx[int(sh[i]):, i][None].contiguous().to(DEV)
        ops.rope_llama_(own, cos[:, :hd // 2].contiguous().to(DEV), sin[:, :hd // 2].contiguous().to(DEV), heads + kvh, hd)
        assert torch.equal(r_tm[int(sh[i]):, i], own.cpu()[0]), i
    lg = torch.randn(5, 4099, generator=g)
    lg[2, 17] = lg[2, 4000] = 9.0                                                  # a tie: the lowest index wins, as torch.argmax
    assert ops.argmax_rows(lg.to(DEV)).cpu().tolist() == [int(x) for x in torch.argmax(lg, 1)] and int(torch.argmax(lg, 1)[2]) == 17
    gu = torch.randn(5, 33, 2 * 1024, generator=g).half()
    sw = ops.swiglu(gu.to(DEV)).float().cpu()
    assert _rel(sw, torch.nn.functional.silu(gu[..., :1024].float()) * gu[..., 1024:].float()) < 2e-3
    hcat = torch.randn(3, 50, 96, generator=g)
    ln = torch.tensor([50, 1, 17])
    mp = ops.mean_pool(hcat.to(DEV), ln.to(DEV, torch.int32)).cpu()
    refp = torch.stack([hcat[i, :int(ln[i])].mean(0) for i in range(3)])
    assert _rel(mp, refp) < 1e-6


def test_embedder_call_surface_and_combined_query():
    """get_embedding / get_embeddings / combined_embedding / generate_emotion_label with the stand-in tokenizer: shapes,
    dtypes, batch == one-at-a-time, and the 2 x hidden concatenation order of src/search_milvus.py:220-221."""
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    cfg = LlamaShape.tiny()
    emb = LlamaEmbedder(make_llama_weights(cfg, 1), cfg, DEV)
    texts = ["I did it, I asked her to marry me.", "neutral", "Elizabeth is a deeply emotional and passionate individual, extremely devoted"]
    one = np.stack([emb.get_embedding(t) for t in texts])
    many = emb.get_embeddings(texts)
    assert one.shape == (3, cfg.hidden) and one.dtype == np.float32
    assert _rel(many, one) < 2e-3
    q = emb.combined_embedding(texts[1], texts[2])
    assert q.shape == (2 * cfg.hidden,) and q.dtype == np.float32
    assert _rel(q[:cfg.hidden], one[1]) < 2e-3 and _rel(q[cfg.hidden:], one[2]) < 2e-3
    label = emb.generate_emotion_label(texts[0], max_new_tokens=3)
    assert isinstance(label, str) and label == label.strip().lower() and len(label) > 0
    long_text = " ".join(["word"] * 2000)
    assert emb.get_embedding(long_text).shape == (cfg.hidden,)                    # truncated to max_length = 512 tokens


def test_search_milvus_cli_end_to_end(capsys):
    """src/search_milvus.py:156-262 restated (astts.cli.search_milvus): text -> GPU embedder (3072-d) x 2 -> 6144-d query ->
    COSINE search on the shipped 130-row bank.  The embedder is the real GPU path at Llama-3.2-3B's widths (3 layers, seeded
    weights); the hits must equal a direct search with the same combined vector, and the printed lines follow the reference."""
    from astts.cli import search_milvus as drv
    from astts.compat.pymilvus import MilvusClient
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    cfg = LlamaShape.wide()
    emb = LlamaEmbedder(make_llama_weights(cfg, 8), cfg, DEV)
    db = os.path.join(GOLD, "milvus_demo.db")
    args = drv.build_parser().parse_args(["--db_path", db, "--search_text", "I did it, I asked her to marry me.", "--query_speaker", "john", "--top_k", "3"])
    res = drv.main(args, embedder=emb)
    out = capsys.readouterr().out
    assert "Generated combined embedding of shape 6144." in out
    assert "Top 3 results for the query 'I did it, I asked her to marry me.':" in out and out.count("File ID: ") == 3
    q = np.concatenate((emb.get_embedding(args.search_text), emb.get_embedding(drv.speaker_bio["JOHN"]))).astype(np.float32)
    direct = MilvusClient(db).search(collection_name=args.collection_name, data=[q.tolist()], anns_field="vector", metric_type="COSINE",
                                     limit=3, output_fields=["file_id"])
    assert [h["row"] for h in res[0]] == [h["row"] for h in direct[0]]
    # unknown speaker -> the literal "unknown" biography (:121); missing collection -> message, no exception (:177-179)
    assert drv.emb_text_bio("nobody", emb).shape == (cfg.hidden,)
    args2 = drv.build_parser().parse_args(["--db_path", db, "--collection_name", "nope"])
    assert drv.main(args2, embedder=emb) is None and "does not exist" in capsys.readouterr().out
    # without a checkpoint the loader refuses unless random-init is explicitly allowed
    old = os.environ.pop("ASTTS_ALLOW_RANDOM_INIT", None)
    try:
        with pytest.raises(FileNotFoundError):
            drv.load_embedder("/nonexistent/llama")
    finally:
        if old is not None:
            os.environ["ASTTS_ALLOW_RANDOM_INIT"] = old


def test_emotion_label_decodes_without_special_tokens_and_with_the_untruncated_prompt():
    """milvus/search_json.py:178-191: the generation prompt is encoded WITHOUT truncation (only get_embedding truncates to 512)
    and the continuation is decoded with skip_special_tokens=True.  A recording tokenizer stands in for the Llama tokenizer."""
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    cfg = LlamaShape.tiny()

    class Tok:
        def __init__(self):
            self.decoded = None

        def encode(self, text):
            return [cfg.bos_token_id] + [3 + (hash(w) % (cfg.vocab - 3)) for w in text.split()]

        def decode(self, ids, skip_special_tokens=False):
            self.decoded = (list(ids), skip_special_tokens)
            return " Happy " if skip_special_tokens else "<|begin_of_text|> Happy <|end_of_text|>"

    tok = Tok()
    emb = LlamaEmbedder(make_llama_weights(cfg, 7), cfg, device=DEV, tokenizer=tok, max_length=16)
    text = " ".join(f"w{i}" for i in range(40))                      # prompt far beyond max_length = 16 and the first RoPE table
    label = emb.generate_emotion_label(text, max_new_tokens=2)
    assert label == "happy"
    ids, skipped = tok.decoded
    assert skipped is True
    assert len(ids) >= len(tok.encode(emb.EMOTION_PROMPT.format(text, text)))     # nothing was cut off the prompt
    assert emb.get_embedding(text).shape == (cfg.hidden,)                          # the embedding path still truncates and works


def test_search_json_text_to_style_ids_in_one_command(tmp_path, capsys):
    """milvus/search_json.py:372-461 restated with its LLM half (astts.cli.search_json --model_path ...): per row emotion label (batched
    KV-cached greedy decode) -> [label | biography] embedding -> top-1 COSINE search -> JSONL.  The JSONL must equal what the same
    driver writes from the precomputed query vectors (--query_npy: the form rounds 1-4 had), the labels must equal the one-at-a-time
    labels, and speakers without a biography get the reference's placeholder text."""
    import json

    from astts.cli import search_json as drv
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    cfg = LlamaShape.wide()
    emb = LlamaEmbedder(make_llama_weights(cfg, 8), cfg, DEV)
    rows = [{"zh_text": "I did it, I asked her to marry me.", "speaker": "m1"}, {"zh_text": "Yeah.", "speaker": "w1"},
            {"zh_text": "", "speaker": "w1"},                                            # skipped, as the reference skips it
            {"zh_text": "Oh my god, that is wonderful news, congratulations to both of you!", "speaker": "w2"},
            {"zh_text": "I don't know. I just feel like nothing is ever going to change.", "speaker": "m1"}]
    inp = tmp_path / "in.jsonl"
    inp.write_text("\n".join(json.dumps(r) for r in rows) + "\n")
    bios = tmp_path / "bios.json"
    bios.write_text(json.dumps({"m1": "A pragmatic and thoughtful person who values honesty.", "w1": "Cheerful and optimistic, always looking for the silver lining."}))
    db = os.path.join(GOLD, "milvus_demo.db")
    out1, out2 = tmp_path / "a.jsonl", tmp_path / "b.jsonl"
    args = drv.build_parser().parse_args(["--input_json", str(inp), "--db_path", db, "--output_file", str(out1), "--biography_json", str(bios),
                                          "--file_prefix_path", "/data/seg_wav", "--llm_batch", "3"])
    res = drv.main(args, embedder=emb)
    assert len(res) == 4 and all(r["retrieved_file_id"].startswith("/data/seg_wav/") for r in res)
    kept = [r for r in rows if r["zh_text"].strip()]
    q, labels, failed = drv.embed_rows(kept, emb, drv.load_biographies(str(bios)), batch=3)
    assert not failed.any()
    assert labels == [emb.generate_emotion_label(r["zh_text"]) for r in kept]            # batched labels == one at a time
    h = cfg.hidden
    assert _rel(q[2, h:], emb.get_embedding(drv.PLACEHOLDER_BIOGRAPHY)) < 2e-3           # w2 has no biography: the reference's fallback text
    assert _rel(q[0, h:], emb.get_embedding("A pragmatic and thoughtful person who values honesty.")) < 2e-3
    assert _rel(q[0, :h], emb.get_embedding(labels[0])) < 2e-3
    full = np.zeros((len(rows), 2 * h), np.float32)
    full[[0, 1, 3, 4]] = q
    np.save(tmp_path / "q.npy", full)
    args2 = drv.build_parser().parse_args(["--input_json", str(inp), "--db_path", db, "--output_file", str(out2), "--query_npy", str(tmp_path / "q.npy"),
                                           "--file_prefix_path", "/data/seg_wav"])
    drv.main(args2)
    assert out1.read_text() == out2.read_text()
    capsys.readouterr()


def test_embedder_at_full_depth_matches_transformers_fixture():
    """The model the reference runs (src/search_milvus.py:75-108, milvus/search_json.py:154-198): all 28 layers of Llama-3.2-3B at its
    real widths over its 128 256-entry vocabulary -- 3.2 B seeded parameters, fp32 transformers on the build container's CPU
    (tests/golden/make_llama_fixtures.py --3b -> llama_3b.npz).  What 28 layers of fp16 MFMA operands do to the residual stream is
    measured layer by layer (printed) and held at the end.  Observed on MI355X (seeded Gaussian weights, which amplify a perturbation
    from layer to layer more than a trained network does): the last token's residual stream 1.7e-3 after one layer, 4.1e-3 after 4,
    1.1e-2 after 16, 1.8e-2 after 28; final hidden states 2.2e-2, mean-pooled embedding 8.9e-3 (cosine to the fixture's embedding
    0.99992: what the retrieval sees), logits over the 128 256-entry vocabulary 1.6e-2.  Bars at ~2x those: hidden <= 4e-2,
    embedding <= 2e-2, logits <= 3e-2, cosine >= 0.9998; the greedy continuation equal."""
    import time
    from astts import ops
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    fx = np.load(os.path.join(GOLD, "llama_3b.npz"))
    cfg = LlamaShape.llama32_3b()
    assert cfg.layers == 28 and cfg.vocab == 128256
    t0 = time.time()
    sd = make_llama_weights(cfg, int(fx["seed"]))
    t1 = time.time()
    emb = LlamaEmbedder(sd, cfg, DEV)
    del sd
    print(f"[3b] weights drawn in {t1 - t0:.1f} s, packed in {time.time() - t1:.1f} s; {torch.cuda.memory_allocated() / 2**30:.1f} GiB resident")
    ids, lens = torch.from_numpy(fx["ids"]), fx["lens"]
    row0 = ids[:1, :lens[0]]
    # error growth by depth: the last token's residual stream after every layer
    x = ops.embedding(emb.embed, row0.to(DEV))
    by_layer = fx["hidden_by_layer_last_token_row0"]
    growth = [_rel(x[0, -1].cpu().numpy(), by_layer[0])]
    cos, sin = emb._rope
    hq, hk = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
    for li, L in enumerate(emb.L):
        h = ops.rmsnorm(x, L["n1"], cfg.rms_eps)
        qkv = ops.linear(h, L["wqkv"], out_dtype=torch.float16)
        ops.rope_llama_(qkv, cos, sin, cfg.heads + cfg.kv_heads, cfg.head_dim)
        a = ops.attn_gqa(qkv[..., :hq], qkv[..., hq:hq + hk], qkv[..., hq + hk:], cfg.heads, cfg.kv_heads, cfg.head_dim)
        x = ops.linear(a, L["wo"], residual=x)
        h = ops.rmsnorm(x, L["n2"], cfg.rms_eps)
        x = ops.linear(ops.swiglu(ops.linear(h, L["wgu"], out_dtype=torch.float16)), L["wd"], residual=x)
        if li + 1 < cfg.layers:                       # (transformers' last entry is taken after the final norm)
            growth.append(_rel(x[0, -1].cpu().numpy(), by_layer[li + 1]))
    print("[3b] residual-stream rel err of the last token after layers 0, 1, 2, 4, 8, 16, 27: " +
          ", ".join(f"{growth[i]:.1e}" for i in (0, 1, 2, 4, 8, 16, 27)))
    h = emb.hidden(row0).cpu().numpy()[0]
    e_h = _rel(h, fx["hidden_final_row0"])
    one = np.stack([emb.embed_ids(ids[i:i + 1, :n]).cpu().numpy()[0] for i, n in enumerate(lens)])
    e_1 = _rel(one, fx["embedding"])
    batched = emb.embed_ids(ids, torch.from_numpy(lens)).cpu().numpy()
    e_b = _rel(batched, fx["embedding"])
    lg = emb.logits_last(row0.to(DEV)).cpu().numpy()[0]
    e_l = _rel(lg, fx["logits_last_row0"])
    # what the retrieval sees: the cosine between this embedding and the fixture's
    cosv = [float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))) for a, b in zip(one.astype(np.float64), fx["embedding"].astype(np.float64))]
    print(f"[parity] llama 3b (28 layers, vocab 128256): final hidden {e_h:.2e}, embedding {e_1:.2e} (one at a time) {e_b:.2e} (padded batch), "
          f"logits {e_l:.2e}; cosine to the fixture's embeddings {min(cosv):.7f}")
    assert e_h < 4e-2 and e_1 < 2e-2 and e_b < 2e-2 and e_l < 3e-2 and min(cosv) > 0.9998
    assert max(growth) < 4e-2 and growth[1] < 5e-3
    n_new = len(fx["greedy"]) - int(lens[0])
    ref = fx["greedy"].tolist()
    gen = emb.generate_greedy(row0[0].tolist(), n_new)
    top2 = np.sort(fx["logits_last_row0"])[-2:]
    print(f"[3b] greedy continuation {gen[-n_new:]} (fixture {ref[-n_new:]}); the fixture's first-step top-2 logit gap {float(top2[1] - top2[0]):.3f}")
    assert gen == ref
