"""CPU tests: pin the kNN oracle against the reference's data artefacts.

The expected values in this file come from two independent places:
  * constants quoted from the survey's probe of the reference DB (SHA-256, top-4 lists, score ranges)
  * tests/golden/* produced by tests/golden/make_fixtures.py from /root/reference data files
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import knn as oknn

# fp64 cosine, stable sort, row-index tie-break -- rows/scores quoted verbatim
SURVEY_TOP4 = {
    0: ([0, 5, 2, 15], [1.0, .954075, .953645, .952324]),
    1: ([1, 5, 2, 0], [1.0, .950185, .949248, .944891]),
    29: ([29, 41, 40, 35], [1.0, .909887, .899838, .896999]),
    61: ([61, 69, 70, 64], [1.0, .960324, .952549, .952188]),
    103: ([103, 89, 106, 104], [1.0, .917591, .906744, .905125]),
    129: ([129, 128, 115, 127], [1.0, .968514, .939389, .939213]),
}


def test_bank_fixture_checksums(real_bank, golden_dir):
    assert real_bank.shape == (130, 6144) and real_bank.dtype == np.float16
    sha16 = hashlib.sha256(real_bank.astype("<f2").tobytes()).hexdigest()
    sha32 = hashlib.sha256(real_bank.astype("<f4").tobytes()).hexdigest()
    assert sha16 == "cc2dc0b8cfc7a386954726c11e0bdee017b8985daba73ec790c76431dc277c74"
    assert sha32 == "5c0a7bac58890c26836193749fa68a150a997392a47c2ad896f9db0c8a77b81a"
    norms = np.linalg.norm(real_bank.astype(np.float64), axis=1)
    assert 35.2 < norms.min() < 35.3 and 43.1 < norms.max() < 43.2
    meta = json.load(open(os.path.join(golden_dir, "style_bank_meta.json")))
    assert meta["metric_type"] == "COSINE" and meta["dim"] == 6144 and meta["n"] == 130
    assert len({m["file_id"] for m in meta["rows"]}) >= 100  # style ids are file ids
    assert len(set(meta["pk"])) < 130  # pk restarts per speaker: not unique (RAG.py:507)


def test_self_retrieval_is_identity(real_bank):
    # the reference's own verification loop: milvus/RAG.py:568-582
    idx, sc = oknn.knn_search(real_bank, real_bank.astype(np.float32), k=1)
    assert np.array_equal(idx[:, 0], np.arange(130))
    assert np.all(np.abs(sc[:, 0] - 1.0) < 1e-12)


def test_survey_known_answers(real_bank):
    q = real_bank.astype(np.float32)
    idx, sc = oknn.knn_search(real_bank, q, k=4)
    for row, (eidx, esc) in SURVEY_TOP4.items():
        assert idx[row].tolist() == eidx
        assert np.allclose(sc[row], esc, atol=5e-7)
    # nearest neighbour excluding self: min .8546, median .9319, max .9895
    nn = sc[:, 1]
    assert abs(nn.min() - .8546) < 1e-4 and abs(np.median(nn) - .9319) < 1e-4 and abs(nn.max() - .9895) < 1e-4


def test_golden_kats_reproduce(real_bank, kats):
    q = real_bank.astype(np.float32)
    idx, sc = oknn.knn_search(real_bank, q, k=5)
    assert idx.tolist() == kats["self_top5_idx"]
    assert np.array_equal(sc, np.asarray(kats["self_top5_score"]))
    gaps = -np.diff(sc, axis=1)
    assert gaps.min() > 0 and gaps.min() < 1e-5  # 4.88e-6: fp32-only ranking is fragile
    qb = q.copy()
    qb[:, :3072] = 0  # biography-only ablation (search_json_ab_bio.py:412)
    bi, _ = oknn.knn_search(real_bank, qb, k=3)
    assert bi.tolist() == kats["bio_only_top3_idx"]
    assert 0.02 < (bi[:, 0] == np.arange(130)).mean() < 0.15  # only ~6 % retrieve themselves


def test_recorded_retrievals_resolve_to_bank(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "style_bank_meta.json")))
    by_id = {m["file_id"]: m["text"] for m in meta["rows"]}
    rows = [json.loads(l) for l in open(os.path.join(golden_dir, "search_results.jsonl"), encoding="utf-8")]
    assert len(rows) == 64
    for r in rows:
        fid = os.path.basename(r["retrieved_file_id"])
        assert fid in by_id and by_id[fid] == r["retrieved_text"]
        assert 0.81 <= r["distance"] <= 0.95  # cosine similarity, larger = closer
        assert set(r) == {"zh_text", "speaker", "retrieved_file_id", "retrieved_text", "distance", "whisper"}


def test_total_order_and_edges():
    rng = np.random.default_rng(0)
    bank = rng.standard_normal((40, 64)).astype(np.float16)
    bank[7] = bank[3]          # exact duplicates -> lower row index first
    bank[21] = bank[3]
    bank[30] = 0               # zero row -> similarity 0
    q = bank[3].astype(np.float32)
    idx, sc = oknn.knn_search(bank, q, k=5)
    assert idx[0, :3].tolist() == [3, 7, 21] and sc[0, 0] == sc[0, 1] == sc[0, 2]
    idx, sc = oknn.knn_search(bank, q, k=100)  # k > n clamps
    assert idx.shape == (1, 40)
    z = np.zeros(64, np.float32)  # zero query: all similarities 0, order = row order
    idx, sc = oknn.knn_search(bank, z, k=4)
    assert idx[0].tolist() == [0, 1, 2, 3] and np.all(sc == 0)
    # scale invariance of the cosine
    a, _ = oknn.knn_search(bank, q * 1000.0, k=10)
    b, _ = oknn.knn_search(bank, q / 1000.0, k=10)
    assert np.array_equal(a, b)


def test_fast_cpu_baseline_agrees_with_oracle(real_bank):
    rng = np.random.default_rng(1)
    b32 = real_bank.astype(np.float32)
    inv = (1.0 / np.linalg.norm(b32.astype(np.float64), axis=1)).astype(np.float32)
    q = b32[:16] + 0.5 * rng.standard_normal((16, 6144)).astype(np.float32)
    i0, s0 = oknn.knn_search(real_bank, q, k=3)
    i1, s1 = oknn.knn_search_fast_f32(b32, inv, q, k=3)
    assert np.array_equal(i0, i1) and np.allclose(s0, s1, atol=1e-12)
