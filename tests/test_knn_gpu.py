"""GPU parity tests for the style-bank kNN: HIP path (through the C ABI) vs oracle/knn.py.

Bar: retrieved ids BIT-EXACT; scores within 1e-6 (the kernel returns the fp64 cosine rounded to
fp32; the oracle's fp64 value may differ in the last fp64 ulps by summation order).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import knn as oknn

pytestmark = pytest.mark.gpu

SCORE_ATOL = 1e-6


def _bank(vectors):
    from astts.knn import StyleBank

    return StyleBank(vectors)


def _check(bank_np, q, k, force_exact=False, sb=None):
    sb = sb or _bank(bank_np)
    idx, sc = sb.search(q, k, force_exact=force_exact)
    eidx, esc = oknn.knn_search(bank_np, q, k)
    kk = eidx.shape[1]
    assert np.array_equal(idx[:, :kk], eidx), (idx[:, :kk][idx[:, :kk] != eidx], eidx[idx[:, :kk] != eidx])
    assert np.allclose(sc[:, :kk], esc, atol=SCORE_ATOL, rtol=0)
    if kk < k:
        assert np.all(idx[:, kk:] == -1) and np.all(np.isneginf(sc[:, kk:]))
    return sb


def test_real_bank_known_answers(real_bank, kats):
    sb = _bank(real_bank)
    assert sb.scan_plane_exact
    q = real_bank.astype(np.float32)
    idx, sc = sb.search(q, 5)
    assert idx.tolist() == kats["self_top5_idx"]            # committed golden ids, bit-exact
    assert np.allclose(sc, np.asarray(kats["self_top5_score"]), atol=SCORE_ATOL, rtol=0)
    assert np.array_equal(idx[:, 0], np.arange(130))        # RAG.py:568-582 self-retrieval
    # the reference's default top_k=3 (search_embeddings.py:64) and top-1 (search_json.py:411)
    for k in (1, 3):
        i2, _ = sb.search(q, k)
        assert i2.tolist() == [r[:k] for r in kats["self_top5_idx"]]


def test_real_bank_ablation_queries(real_bank, kats):
    sb = _bank(real_bank)
    q = real_bank.astype(np.float32)
    qb = q.copy()
    qb[:, :3072] = 0                                        # search_json_ab_bio.py:412 (adversarial near-ties)
    idx, sc = sb.search(qb, 3)
    assert idx.tolist() == kats["bio_only_top3_idx"]
    assert np.allclose(sc, np.asarray(kats["bio_only_top3_score"]), atol=SCORE_ATOL, rtol=0)
    qe = q.copy()
    qe[:, 3072:] = 0
    qe /= np.linalg.norm(qe, axis=1, keepdims=True)         # search_json_ab_text.py:412,420 (L2-normalised)
    idx, sc = sb.search(qe.astype(np.float32), 3)
    assert idx.tolist() == kats["emo_only_norm_top3_idx"]
    assert np.allclose(sc, np.asarray(kats["emo_only_norm_top3_score"]), atol=SCORE_ATOL, rtol=0)


def test_fp32_upload_of_exact_bank_matches(real_bank):
    sb = _bank(real_bank.astype(np.float32))                # FloatVector semantics: fp32 in, fp16-exact detected
    assert sb.scan_plane_exact
    _check(real_bank, real_bank[:16].astype(np.float32) * 3.0, 3, sb=sb)


@pytest.mark.parametrize("n,d,nq,k", [(1000, 6144, 8, 3), (1000, 6144, 64, 3), (777, 768, 33, 5),
                                      (5000, 768, 130, 1), (20000, 256, 256, 3), (300, 100, 7, 8),
                                      (64, 64, 5, 16), (2000, 6144, 300, 3)])
def test_synthetic_banks_match_oracle(n, d, nq, k, real_bank):
    rng = np.random.default_rng(1234)
    if (n, d) == (1000, 6144):
        # BASELINE config 2: real rows tiled + perturbed, re-rounded to fp16
        bank = (real_bank[np.arange(n) % 130].astype(np.float32)
                + 0.05 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float16)
        rq = np.random.default_rng(0)
        q = bank[rq.integers(0, n, nq)].astype(np.float32) + 0.5 * rq.standard_normal((nq, d)).astype(np.float32)
    else:
        bank = rng.standard_normal((n, d)).astype(np.float16)
        q = rng.standard_normal((nq, d)).astype(np.float32)
    sb = _check(bank, q, k)
    assert sb.last_fallbacks() <= nq // 4                    # the certified fast path carries the load


def test_exact_path_forced_matches_oracle():
    rng = np.random.default_rng(7)
    bank = rng.standard_normal((3000, 512)).astype(np.float16)
    q = rng.standard_normal((20, 512)).astype(np.float32)
    sb = _check(bank, q, 4, force_exact=True)
    assert sb.last_fallbacks() == 20
    _check(bank, q, 4, force_exact=False, sb=sb)


def test_inexact_fp32_bank():
    rng = np.random.default_rng(3)
    bank = rng.standard_normal((1500, 384)).astype(np.float32)  # not fp16-representable
    q = rng.standard_normal((9, 384)).astype(np.float32)
    sb = _bank(bank)
    assert not sb.scan_plane_exact
    _check(bank, q, 3, sb=sb)
    _check(bank, q, 3, force_exact=True, sb=sb)


def test_fp32_bank_rows_of_any_norm():
    """cosine is scale invariant, but an fp32 bank row of norm ~1e-3 (elements ~1e-5, below fp16's normal
    range) used to go subnormal / flush to zero in the fp16 scan planes and drop out of the candidate lists while the
    query was still certified.  The approximate planes of an inexact bank now carry a power-of-two scale per row.  Rows
    with norms from 1e-6 to 1e+7 (values beyond fp16's range included), queried at their own scale and at unit scale,
    through the register scan (Q=9) and the GEMM scan (Q=70)."""
    rng = np.random.default_rng(11)
    n, d = 9000, 320
    base = rng.standard_normal((n, d)).astype(np.float32)
    norms = 10.0 ** rng.uniform(-6, 7, size=n)
    bank = (base * (norms / np.linalg.norm(base, axis=1))[:, None]).astype(np.float32)
    sb = _bank(bank)
    assert not sb.scan_plane_exact
    pick = rng.integers(0, n, 70)
    noise = 0.3 * rng.standard_normal((70, d)).astype(np.float32)
    q_own = bank[pick] * (1.0 + noise)                                  # near its row, at the row's own (tiny or huge) scale
    q_unit = base[pick] / np.linalg.norm(base[pick], axis=1)[:, None] * (1.0 + noise)
    for q in (q_own, q_unit):
        for sl in (slice(0, 9), slice(0, 70)):
            idx, _ = sb.search(q[sl], 3)
            assert np.array_equal(idx[:, 0], pick[sl])                  # every row is found, whatever its norm
            _check(bank, q[sl], 3, sb=sb)
    assert sb.last_fallbacks() <= 2                                     # ... and by the certified fast path, not the exact rescan


def test_gemm_scan_path_large_query_groups():
    """Query groups of >= 64 against a bank that fills the chip take the scan as one GEMM on the ring kernel: exact and
    fp16-inexact banks, a tail group below 64 queries, duplicate rows (ties decided by row index), scaled queries."""
    rng = np.random.default_rng(11)
    bank16 = rng.standard_normal((20000, 256)).astype(np.float16)
    bank16[777] = bank16[123]                       # exact duplicates
    bank16[19999] = bank16[123]
    q = rng.standard_normal((300, 256)).astype(np.float32)       # groups of 256 + 44 (tail: register-streaming scan)
    q[5] = bank16[123].astype(np.float32) * 37.5
    q[260] = bank16[777].astype(np.float32) * 1e-3
    sb = _check(bank16, q, 5)
    assert sb.last_fallbacks() <= 75
    _check(bank16, q[:70], 3, force_exact=True, sb=sb)
    bank32 = rng.standard_normal((12000, 384)).astype(np.float32)  # not fp16-representable: the scan plane is a rounded image
    sb32 = _bank(bank32)
    assert not sb32.scan_plane_exact
    _check(bank32, rng.standard_normal((130, 384)).astype(np.float32), 3, sb=sb32)


def test_large_query_group_selection_on_ordered_rows():
    """Query groups of >= 64 against a long score row take one of two selections: from the block maxima the GEMM scan leaves beside
    its scores (knn_blocks_rescore: unmasked single-pass searches), or one streaming block per query against the c-th best so far
    (knn_select_stream: searches with a row mask -- here a mask of ones -- and the passes of limit > 32).  Scores that ASCEND with the row
    index make every score a survivor of the stream (its pool overflows in every span: the tile-by-tile histogram selection takes over),
    scores that DESCEND leave none after the first tile, a bank of seven distinct rows repeated is all ties (decided by row index, across
    blocks), and n is no multiple of 64 -- each against the oracle, with 3 and with 20 hits (the 16- and 64-entry lists)."""
    rng = np.random.default_rng(3)
    n, d = 40013, 64
    u = rng.standard_normal(d)
    u /= np.linalg.norm(u)
    v = rng.standard_normal(d)
    v -= (v @ u) * u
    v /= np.linalg.norm(v)
    t = np.linspace(2.0, 0.05, n)[:, None]                   # angle to u shrinks with the row index: cosine ascends
    asc = (u[None, :] + t * v[None, :]).astype(np.float32) * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    q = (u[None, :] + 0.01 * rng.standard_normal((64, d))).astype(np.float32)
    ones = np.ones(n, np.uint8)
    for bank in (asc, asc[::-1].copy()):
        sb = _check(bank, q, 3)
        _check(bank, q, 20, sb=sb)
        _check_metric(bank, q, 3, "COSINE", mask=ones, sb=sb)
        _check_metric(bank, q, 20, "COSINE", mask=ones, sb=sb)
    few = rng.standard_normal((7, d)).astype(np.float16)
    ties = few[rng.integers(0, 7, n)]
    qt = rng.standard_normal((70, d)).astype(np.float32)
    sb = _check(ties, qt, 5)
    _check(ties, few.astype(np.float32).repeat(10, 0), 20, sb=sb)
    _check_metric(ties, qt, 5, "COSINE", mask=ones, sb=sb)
    for metric in ("IP", "L2"):                              # the epilogue's per-row constant (L2) and the plain scale (IP), ties included
        sbm = _check_metric(ties, qt, 5, metric)
        _check_metric(ties, qt, 20, metric, mask=ones, sb=sbm)


def test_large_query_group_beyond_the_block_maximum_range():
    """More than 8192 blocks of 64 rows (n > 524 288): the block maxima no longer fit one selection segment, so an unmasked search of a
    large query group streams the row as a masked one does; the shortest GEMM-scanned row (two segments, n just above 8192) and a query
    count that is cut into equal groups (300 = 150 + 150) ride along."""
    rng = np.random.default_rng(17)
    bank = rng.standard_normal((530000, 64)).astype(np.float16)
    q = bank[rng.integers(0, 530000, 70)].astype(np.float32) + 0.3 * rng.standard_normal((70, 64)).astype(np.float32)
    sb = _check(bank, q, 3)
    _check(bank, q[:64], 20, sb=sb)
    small = rng.standard_normal((8200, 128)).astype(np.float16)
    _check(small, rng.standard_normal((300, 128)).astype(np.float32), 5)


def test_duplicates_ties_zero_rows_and_small_banks():
    rng = np.random.default_rng(0)
    bank = rng.standard_normal((200, 128)).astype(np.float16)
    for j in range(40):                       # 40 exact copies of row 3: more than the candidate list holds
        bank[5 * j] = bank[3] if j else bank[0]
    bank[199] = 0                             # zero row
    q = np.stack([bank[3], bank[0], np.zeros(128, np.float16), bank[17]]).astype(np.float32)
    sb = _check(bank, q, 3)
    assert sb.last_fallbacks() >= 1           # uncertifiable ties must go through the exact scan
    _check(bank, q, 20, sb=sb)
    tiny = rng.standard_normal((5, 70)).astype(np.float16)      # n < k, d not a multiple of 64
    _check(tiny, rng.standard_normal((3, 70)).astype(np.float32), 8)
    one = rng.standard_normal((1, 33)).astype(np.float16)
    _check(one, rng.standard_normal((2, 33)).astype(np.float32), 1)


def test_query_scale_extremes():
    rng = np.random.default_rng(5)
    bank = rng.standard_normal((900, 256)).astype(np.float16)
    q = rng.standard_normal((6, 256)).astype(np.float32)
    sb = _bank(bank)
    for s in (1e-30, 1e-8, 1.0, 1e8, 1e30):
        _check(bank, q * np.float32(s), 3, sb=sb)


def test_device_api_is_async_and_reusable():
    rng = np.random.default_rng(11)
    bank = rng.standard_normal((4096, 768)).astype(np.float16)
    sb = _bank(bank)
    qs = [torch.from_numpy(rng.standard_normal((8, 768)).astype(np.float32)).cuda() for _ in range(5)]
    outs = [sb.search_device(q, 3) for q in qs]            # queued back to back, one workspace
    torch.cuda.synchronize()
    for q, (i, s) in zip(qs, outs):
        ei, es = oknn.knn_search(bank, q.cpu().numpy(), 3)
        assert np.array_equal(i.cpu().numpy(), ei)


def test_argument_errors():
    from astts import _lib
    from astts.knn import StyleBank

    sb = StyleBank(np.ones((10, 64), np.float16))
    with pytest.raises(ValueError):
        sb.search(np.ones((1, 63), np.float32), 3)
    with pytest.raises(ValueError):
        sb.search(np.ones((1, 64), np.float32), 0)
    with pytest.raises(ValueError):
        sb.search(np.ones((1, 64), np.float32), _lib.KNN_MAX_K + 1)
    with pytest.raises(_lib.AsttsError):
        StyleBank(np.full((4, 64), np.inf, np.float32))    # non-finite values
    with pytest.raises(ValueError):
        StyleBank(np.ones((4, 64), np.float16), metric="HAMMING")
    with pytest.raises(ValueError):
        sb.search(np.ones((1, 64), np.float32), 3, row_mask=np.ones(9, np.uint8))       # mask of the wrong length


def test_full_size_bank_properties():
    """BASELINE config 5 size (100k x 6144, Q=256): size-independent properties + oracle on a sample."""
    n, d, nq, k = 100_000, 6144, 256, 3
    g = torch.Generator(device="cuda").manual_seed(1234)
    bank_t = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    from astts.knn import StyleBank

    sb = StyleBank(bank_t)
    rows = torch.randint(0, n, (nq,), generator=g, device="cuda")
    q = bank_t[rows].to(torch.float32)
    idx, sc = sb.search_device(q, k)
    torch.cuda.synchronize()
    assert torch.equal(idx[:, 0], rows)                      # self-retrieval at full size
    assert torch.allclose(sc[:, 0], torch.ones(nq, device="cuda"), atol=1e-6)
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())             # sorted
    noisy = q + 0.7 * torch.randn(q.shape, generator=g, device="cuda")
    i_fast, s_fast = sb.search_device(noisy, k)
    i_fast, s_fast = i_fast.clone(), s_fast.clone()
    nfb = sb.last_fallbacks()
    i_ex, s_ex = sb.search_device(noisy, k, force_exact=True)
    assert torch.equal(i_fast, i_ex) and torch.equal(s_fast, s_ex)   # certified fast path == exact scan
    assert nfb <= 8
    bank_np = bank_t.cpu().numpy()
    sample = noisy[:3].cpu().numpy()
    ei, es = oknn.knn_search(bank_np, sample, k)
    assert np.array_equal(i_fast[:3].cpu().numpy(), ei)
    assert np.allclose(s_fast[:3].cpu().numpy(), es, atol=SCORE_ATOL, rtol=0)


def test_milvus_client_search_end_to_end(golden_dir, kats):
    """The reference call, verbatim (milvus/search_embeddings.py:15-22), on the shipped DB."""
    from astts.compat.pymilvus import MilvusClient

    client = MilvusClient(os.path.join(golden_dir, "milvus_demo.db"))
    meta = json.load(open(os.path.join(golden_dir, "style_bank_meta.json")))
    bank = np.load(os.path.join(golden_dir, "style_bank_130x6144.f16.npy")).astype(np.float32)
    embedding = bank[61].tolist()
    results = client.search(collection_name="embeddings_biographies_collection", data=[embedding],
                            anns_field="vector", param={"nprobe": 10}, limit=3,
                            output_fields=["file_id", "text"])
    assert len(results) == 1 and len(results[0]) == 3
    assert [h["row"] for h in results[0]] == kats["self_top5_idx"][61][:3]
    top = results[0][0]
    assert top["entity"] == meta["rows"][61] and top["id"] == meta["pk"][61]
    assert abs(top["distance"] - 1.0) < 1e-6
    assert results[0][0]["distance"] >= results[0][1]["distance"] >= results[0][2]["distance"]
    # src/search_milvus.py:140-147 form, ndarray query (search_json.py:411), top-1
    r2 = client.search(collection_name="embeddings_biographies_collection", data=[bank[5]],
                       anns_field="vector", metric_type="COSINE", limit=1, output_fields=["file_id"])
    assert r2[0][0]["entity"] == {"file_id": meta["rows"][5]["file_id"]}
    client.close()


def test_bank_sharded_shards_on_one_gpu_equal_unsharded():
    """SURVEY.md 8e stress mode, emulated on one GPU: the bank cut into 3 row shards (three StyleBank handles), every shard
    searched with all queries (fp64 scores through astts_knn_search_f64), candidates merged by astts.parallel.merge_topk ->
    identical to the unsharded search and to the oracle, duplicate rows across shards resolved by the lower row."""
    from astts.knn import StyleBank
    from astts.parallel import merge_topk, shard_bounds

    rng = np.random.default_rng(5)
    n, d, nq, k = 4000, 256, 40, 5
    bank = rng.standard_normal((n, d)).astype(np.float16)
    bank[3100] = bank[17]
    q = bank[rng.integers(0, n, nq)].astype(np.float32) + 0.4 * rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = bank[17]
    qd = torch.from_numpy(q).cuda()
    cand_s, cand_r = [], []
    for r in range(3):
        b0, b1, _ = shard_bounds(n, 3, r)
        sb = StyleBank(bank[b0:b1])
        idx, sc32, sc64 = sb.search_device(qd, k, return_f64=True)
        assert sc64.dtype == torch.float64 and torch.equal(sc64.to(torch.float32), sc32)
        cand_s.append(sc64)
        cand_r.append(torch.where(idx >= 0, idx + b0, idx))
    idx, sc = merge_topk(torch.cat(cand_s, 1), torch.cat(cand_r, 1), k)
    ei, es = oknn.knn_search(bank, q, k)
    assert np.array_equal(idx.cpu().numpy(), ei)
    assert np.allclose(sc.cpu().numpy(), es, atol=1e-12, rtol=0)
    assert idx[0, :2].tolist() == [17, 3100]
    full_idx, _ = StyleBank(bank).search_device(qd, k)
    assert torch.equal(full_idx, idx)


# ------------------------------------------------------------------------------------------ IP / L2, row masks, k > 32 (round 6)
METRICS = {"COSINE": oknn.METRIC_COSINE, "IP": oknn.METRIC_IP, "L2": oknn.METRIC_L2}


def _check_metric(bank_np, q, k, metric, mask=None, force_exact=False, sb=None, atol=None):
    from astts.knn import StyleBank

    sb = sb or StyleBank(bank_np, metric=metric)
    idx, sc = sb.search(q, k, force_exact=force_exact, row_mask=mask)
    eidx, esc = oknn.knn_search(bank_np, q, k, METRICS[metric], row_mask=mask)
    kk = eidx.shape[1]
    assert np.array_equal(idx[:, :kk], eidx), (metric, np.argwhere(idx[:, :kk] != eidx)[:5], idx[:, :kk][idx[:, :kk] != eidx][:5], eidx[idx[:, :kk] != eidx][:5])
    live = eidx >= 0
    scale = max(1.0, float(np.abs(esc[live]).max())) if live.any() else 1.0
    assert np.allclose(sc[:, :kk][live], esc[live], atol=(atol or SCORE_ATOL) * scale, rtol=0)
    dead = ~live
    if dead.any():
        assert np.all(np.isposinf(sc[:, :kk][dead]) if metric == "L2" else np.isneginf(sc[:, :kk][dead]))
    if kk < k:
        assert np.all(idx[:, kk:] == -1)
    return sb


@pytest.mark.parametrize("metric", ["IP", "L2"])
def test_ip_and_l2_on_the_real_bank(real_bank, metric):
    """The shipped 130 x 6144 bank under the other two MilvusClient metrics: ids bit-exact against the fp64 oracle, scores to fp32
    rounding; self-queries (L2: distance exactly 0 first), noisy queries, both paths (certified candidates / exact scan)."""
    q = real_bank.astype(np.float32)
    sb = _check_metric(real_bank, q, 5, metric)
    if metric == "L2":
        idx, sc = sb.search(q, 1)
        assert np.array_equal(idx[:, 0], np.arange(130)) and np.all(sc[:, 0] == 0.0)
    rng = np.random.default_rng(3)
    noisy = q[:32] + 0.5 * rng.standard_normal((32, 6144)).astype(np.float32)
    _check_metric(real_bank, noisy, 3, metric, sb=sb)
    _check_metric(real_bank, noisy, 3, metric, sb=sb, force_exact=True)
    _check_metric(real_bank, noisy[:4] * 1e-3, 8, metric, sb=sb)      # tiny queries: the power-of-two pre-scale keeps the fp16 image exact


@pytest.mark.parametrize("metric", ["COSINE", "IP", "L2"])
@pytest.mark.parametrize("n,d,nq", [(1000, 6144, 8), (20000, 768, 96), (9000, 1280, 300)])
def test_metrics_on_synthetic_banks(metric, n, d, nq):
    """fp32 (not fp16-exact) Gaussian banks with rows of very different norms: the K-split scan (small bank), the GEMM scan (>= 64
    queries against a large bank), several selection segments (n > 8192) and several query groups (nq > 256)."""
    rng = np.random.default_rng(n + d)
    bank = (rng.standard_normal((n, d)) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
    q = (bank[rng.integers(0, n, nq)] + 0.7 * rng.standard_normal((nq, d))).astype(np.float32)
    sb = _check_metric(bank, q, 3, metric, atol=2e-6)
    assert sb.last_fallbacks() <= nq
    _check_metric(bank, q[:5], 20, metric, sb=sb, atol=2e-6)           # the 64-entry candidate list


def test_vq_argmin_shape():
    """What the speech tokenizer's quantiser asks for: k = 1, L2, a 4096 x 1280 codebook, a few hundred unit-norm frames."""
    rng = np.random.default_rng(11)
    code = rng.standard_normal((4096, 1280)).astype(np.float32)
    code /= np.linalg.norm(code, axis=1, keepdims=True)
    x = code[rng.integers(0, 4096, 375)] + 0.05 * rng.standard_normal((375, 1280)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    _check_metric(code, x, 1, "L2", atol=2e-6)


@pytest.mark.parametrize("metric", ["COSINE", "L2"])
def test_row_masks(real_bank, metric):
    """A filter's row mask: hits are the exact top-k of the allowed rows -- one mask for every query, one per query, masks that
    leave fewer than k rows (missing hits are -1), an empty mask, and the exact path."""
    q = real_bank[:16].astype(np.float32)
    rng = np.random.default_rng(5)
    shared = (rng.random(130) < 0.5).astype(np.uint8)
    sb = _check_metric(real_bank, q, 5, metric, mask=shared)
    per_q = (rng.random((16, 130)) < 0.3).astype(np.uint8)
    _check_metric(real_bank, q, 5, metric, mask=per_q, sb=sb)
    few = np.zeros(130, np.uint8)
    few[[7, 99]] = 1
    _check_metric(real_bank, q, 5, metric, mask=few, sb=sb)
    _check_metric(real_bank, q, 5, metric, mask=few, sb=sb, force_exact=True)
    _check_metric(real_bank, q, 3, metric, mask=np.zeros(130, np.uint8), sb=sb)
    # a large bank: several selection segments, most rows masked in some of them
    n = 30000
    bank = rng.standard_normal((n, 256)).astype(np.float16)
    big = (rng.random(n) < 0.02).astype(np.uint8)
    big[:9000] = 0
    _check_metric(bank, bank[:40].astype(np.float32) + 0.1, 10, metric, mask=big)


@pytest.mark.parametrize("metric", ["COSINE", "IP", "L2"])
def test_limits_beyond_32_hits(real_bank, metric):
    """limit up to 1024: 32 certified hits per selection pass over ONE scan, the rows of earlier passes masked out -- the
    concatenation is the oracle's order, through every pass boundary; k > n returns every row, then -1."""
    q = real_bank[:6].astype(np.float32) + 0.25
    sb = _check_metric(real_bank, q, 33, metric)
    _check_metric(real_bank, q, 100, metric, sb=sb)
    _check_metric(real_bank, q, 130, metric, sb=sb)
    _check_metric(real_bank, q, 200, metric, sb=sb)                    # more hits than rows
    half = np.zeros(130, np.uint8)
    half[::2] = 1
    _check_metric(real_bank, q, 70, metric, sb=sb, mask=half)          # 65 allowed rows
    rng = np.random.default_rng(8)
    bank = rng.standard_normal((20000, 128)).astype(np.float16)
    qq = rng.standard_normal((300, 128)).astype(np.float32)            # two chunks of <= 256 queries
    _check_metric(bank, qq, 1024 if metric == "COSINE" else 80, metric, atol=2e-6)


def test_milvus_client_filter_limit_and_metrics(tmp_path):
    """The client surface of those three: filter expressions over the dynamic fields and the primary key, limit > 32, and an L2 / IP
    collection (distance ascending / inner product descending), each against the oracle on the rows the filter allows."""
    from astts.compat.pymilvus import MilvusClient, MilvusException

    rng = np.random.default_rng(21)
    n, d = 300, 64
    vecs = rng.standard_normal((n, d)).astype(np.float32)
    rows = [{"id": i, "vector": vecs[i].tolist(), "file_id": f"spk{i % 7}_{i}.wav", "text": "yes" if i % 3 == 0 else "no", "dur": float(i) / 10}
            for i in range(n)]
    q = vecs[5] + 0.1
    for metric in ("COSINE", "L2", "IP"):
        c = MilvusClient(str(tmp_path / f"{metric}.db"))
        c.create_collection(collection_name="bank", dimension=d, metric_type=metric)
        c.insert("bank", rows)
        for flt, allow in (('text == "yes"', [i % 3 == 0 for i in range(n)]),
                           ('file_id like "spk3%" and id >= 100', [i % 7 == 3 and i >= 100 for i in range(n)]),
                           ('$meta["text"] in ["no"] or dur < 1.55', [i % 3 != 0 or i / 10 < 1.55 for i in range(n)]),
                           ('not (id in [5, 6, 7])', [i not in (5, 6, 7) for i in range(n)])):
            hits = c.search("bank", data=[q.tolist()], limit=40, filter=flt, output_fields=["file_id"])[0]
            ei, es = oknn.knn_search(vecs, q[None], 40, METRICS[metric], row_mask=np.asarray(allow))
            want = [int(i) for i in ei[0] if i >= 0]
            assert [h["row"] for h in hits] == want and [h["id"] for h in hits] == want, (metric, flt)
            assert np.allclose([h["distance"] for h in hits], es[0][:len(want)], atol=1e-4)
            assert all(h["entity"]["file_id"] == rows[h["row"]]["file_id"] for h in hits)
        with pytest.raises(MilvusException):
            c.search("bank", data=[q.tolist()], limit=3, filter="text === 3")
        c.close()


@pytest.mark.parametrize("metric", ["COSINE", "IP", "L2"])
def test_small_bank_two_launch_form_at_the_extremes(real_bank, metric):
    """Banks of one selection segment searched with <= 32 queries skip the query-preparation launch: the scan rounds the caller's
    fp32 queries to fp16 as they are (no power-of-two pre-scale).  What that costs must end in the certification bound or in the exact
    path, never in the ids: tiny queries (elements far below fp16's normal range), huge ones (beyond fp16's range: +-inf in the scan's
    image), mixed magnitudes inside one query, a zero query, k larger than the candidate list's fast form."""
    q = real_bank[:12].astype(np.float32)
    sb = _check_metric(real_bank, q * 1e-7, 3, metric)
    _check_metric(real_bank, q * 1e-3, 3, metric, sb=sb)
    _check_metric(real_bank, q * 3e4, 3, metric, sb=sb)                 # elements up to ~1e5: past 65504
    assert sb.last_fallbacks() == 12
    mixed = q.copy() * 1e-6
    mixed[:, ::97] = q[:, ::97] * 50.0
    _check_metric(real_bank, mixed, 3, metric, sb=sb)
    z = np.zeros((2, 6144), np.float32)
    z[1, 5] = 1.0
    _check_metric(real_bank, z, 3, metric, sb=sb)
    _check_metric(real_bank, q, 20, metric, sb=sb)                      # 64-entry candidate list (histogram selection)
    # a plain run certifies (no query takes the exact scan)
    _check_metric(real_bank, q + 0.3, 3, metric, sb=sb)
    assert sb.last_fallbacks() == 0
