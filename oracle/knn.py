"""CPU oracle for style-embedding kNN retrieval.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py for who may import this).

Restates what the reference obtains from ``MilvusClient.search`` on a COSINE
collection -- the arithmetic is in milvus-lite (un-vendored, unpinned), so this
follows Milvus' documented semantics and the reference's call sites:

* /root/reference/milvus/search_embeddings.py:9-27   (search, top_k, output_fields)
* /root/reference/src/search_milvus.py:126-152       (metric_type="COSINE")
* /root/reference/src/search_milvus.py:214-221       (query = concat(emotion, bio), fp32, un-normalised)
* /root/reference/milvus/search_json.py:412-429      (top-1 hit -> retrieval record; `distance` is the
                                                      cosine *similarity*, larger = closer)
* /root/reference/milvus/RAG.py:568-582              (self-retrieval check: top-1 of a bank row is itself)

Definition (total order, so ids are reproducible bit-for-bit):
    score(q, n) = <q, b_n> / (sqrt(<q,q>) * sqrt(<b_n,b_n>))          in IEEE fp64
    hits sorted by (score descending, row index ascending); first k returned.
Row sums are taken with numpy's pairwise summation along the contiguous axis
of ``bank * q`` so that identical rows get bit-identical scores regardless of
their position in the bank (a BLAS gemv does not promise that).

Parity: PINNED against the reference's shipped style bank and recorded
retrievals (tests/golden/, tests/test_oracle_knn.py).
"""
from __future__ import annotations

import numpy as np

METRIC_COSINE = 0
METRIC_IP = 1
METRIC_L2 = 2


def _as_f64_rows(x) -> np.ndarray:
    a = np.asarray(x)
    if a.ndim == 1:
        a = a[None, :]
    return np.ascontiguousarray(a, dtype=np.float64)


def scores_f64(bank, queries, metric: int = METRIC_COSINE, chunk_rows: int = 4096) -> np.ndarray:
    """Exact-as-fp64 score matrix ``[Q, N]``.

    ``bank`` is ``[N, D]`` (fp16 or fp32 values, promoted exactly), ``queries`` ``[Q, D]`` fp32.
    COSINE -> cosine similarity, IP -> inner product, L2 -> *negated* squared distance
    (so that "larger is closer" holds for every metric inside the oracle).
    """
    q64 = _as_f64_rows(queries)
    bank = np.asarray(bank)
    n, d = bank.shape
    assert q64.shape[1] == d, (q64.shape, bank.shape)
    out = np.empty((q64.shape[0], n), dtype=np.float64)
    qn = np.sqrt((q64 * q64).sum(axis=1))
    for r0 in range(0, n, chunk_rows):
        b = np.ascontiguousarray(bank[r0:r0 + chunk_rows], dtype=np.float64)
        bn2 = (b * b).sum(axis=1)
        for qi in range(q64.shape[0]):
            dot = (b * q64[qi]).sum(axis=1)
            if metric == METRIC_COSINE:
                with np.errstate(divide="ignore", invalid="ignore"):
                    s = dot / (qn[qi] * np.sqrt(bn2))
                s = np.where(np.isfinite(s), s, 0.0)  # zero vector -> similarity 0
            elif metric == METRIC_IP:
                s = dot
            elif metric == METRIC_L2:
                df = b - q64[qi]
                s = -(df * df).sum(axis=1)          # the difference first (faiss fvec_L2sqr, what Milvus' FLAT scan calls): exactly 0 for a row equal to the query
            else:
                raise ValueError(f"unknown metric {metric}")
            out[qi, r0:r0 + b.shape[0]] = s
    return out


def topk_from_scores(scores: np.ndarray, k: int):
    """Stable (score desc, index asc) top-k.  Returns (idx int64 [Q,k'], score f64 [Q,k'])."""
    qn, n = scores.shape
    k = min(k, n)
    idx = np.empty((qn, k), dtype=np.int64)
    val = np.empty((qn, k), dtype=np.float64)
    rows = np.arange(n)
    for qi in range(qn):
        order = np.lexsort((rows, -scores[qi]))  # primary: -score asc; secondary: row asc
        idx[qi] = order[:k]
        val[qi] = scores[qi, order[:k]]
    return idx, val


def knn_search(bank, queries, k: int, metric: int = METRIC_COSINE, row_mask=None):
    """Oracle top-k.  Returns (idx int64 [Q,k], score float64 [Q,k]); the score is the metric's own value (COSINE: similarity,
    IP: inner product, L2: SQUARED distance -- Milvus' convention), hits closest first, ties by row index ascending.
    ``row_mask`` ([N] or [Q, N], non-zero = allowed): a Milvus ``filter`` -- rows outside it cannot be hits; missing hits are
    (-1, -inf) (+inf for L2)."""
    sc = scores_f64(bank, queries, metric)
    if row_mask is not None:
        m = np.broadcast_to(np.asarray(row_mask) != 0, sc.shape)
        sc = np.where(m, sc, -np.inf)
    idx, val = topk_from_scores(sc, k)
    if row_mask is not None:
        dead = ~np.isfinite(val) & (val < 0)
        idx = np.where(dead, -1, idx)
    if metric == METRIC_L2:
        val = -val
    return idx, val


def knn_search_fast_f32(bank_f32: np.ndarray, inv_norm: np.ndarray, queries, k: int):
    """The *CPU baseline* leg for bench.py: what a tuned CPU implementation of the same
    job does -- one fp32 SGEMM against the resident bank, top-(k+16) by argpartition,
    fp64 re-score of those candidates, stable order.  Not used for parity (ids are
    checked against :func:`knn_search`), only timed.
    """
    q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries, dtype=np.float32)))
    s = (q @ bank_f32.T) * inv_norm[None, :]
    c = min(k + 16, bank_f32.shape[0])
    cand = np.argpartition(-s, c - 1, axis=1)[:, :c]
    idx = np.empty((q.shape[0], min(k, c)), dtype=np.int64)
    val = np.empty(idx.shape, dtype=np.float64)
    for qi in range(q.shape[0]):
        rows = np.sort(cand[qi])
        sc = scores_f64(bank_f32[rows], q[qi])[0]
        order = np.lexsort((rows, -sc))[: idx.shape[1]]
        idx[qi] = rows[order]
        val[qi] = sc[order]
    return idx, val
