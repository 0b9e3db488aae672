"""Data-parallel sharding of utterance / query batches across the GPUs of one node.

The reference processes utterances one by one with no cross-item state
(/root/reference/tts_with_rag.py:172-197, /root/reference/milvus/search_json.py:382-411), so the
path shards by item: rank r takes items [r*ceil(Q/W), (r+1)*ceil(Q/W)); the style bank and the model
weights are replicated.  The ONLY collective on the data path is one all-gather of the retrieved
style ids (int64 [ceil(Q/W), k] per rank, a few KB: latency-bound on xGMI) -- no waveform or mel
crosses GPUs.  One process per GPU; backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in CPU tests.

Stress mode (BASELINE config 5, SURVEY.md 8e "optional"): the BANK is sharded instead -- rank r holds rows
[r*ceil(N/W), (r+1)*ceil(N/W)), every rank sees every query, returns the top-k of its rows, and one all-gather of the
[Q, k] (fp64 score, global row) pairs is followed by a local k-way merge in the oracle's total order (score descending, row
ascending): ``bank_sharded_search``.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


RANK_FAILED = -(1 << 62)     # an id no bank row has: a rank whose local search raised sends it in place of its rows (sharded_search)


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int, int]:
    """-> (begin, end, per_rank) with per_rank = ceil(n/world); trailing ranks may be short/empty."""
    per = (n_items + world - 1) // world if n_items > 0 else 0
    b = min(rank * per, n_items)
    e = min(b + per, n_items)
    return b, e, per


def gather_style_ids(local_ids: torch.Tensor, dist, pad_rows: Optional[int] = None, pad_value: int = -1) -> torch.Tensor:
    """All-gather ``local_ids`` ([q_local, k] int64) from every rank -> [W * pad_rows, k].
    Every rank must contribute ``pad_rows`` rows (short shards are padded with ``pad_value``)."""
    world = dist.get_world_size()
    rows = pad_rows if pad_rows is not None else local_ids.shape[0]
    if local_ids.shape[0] != rows:
        pad = torch.full((rows - local_ids.shape[0], local_ids.shape[1]), pad_value,
                         dtype=local_ids.dtype, device=local_ids.device)
        local_ids = torch.cat([local_ids, pad], dim=0)
    out = torch.empty((world * rows, local_ids.shape[1]), dtype=local_ids.dtype, device=local_ids.device)
    dist.all_gather_into_tensor(out, local_ids.contiguous())
    return out


def sharded_search(search_fn: Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]],
                   queries: torch.Tensor, k: int, dist=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Query-sharded retrieval.  ``queries`` is the FULL [Q, D] batch (same on every rank);
    ``search_fn(q_local, k) -> (idx [q,k] int64, score [q,k] fp32)`` is this rank's bank search
    (StyleBank.search_device on GPUs).  Returns the full (idx [Q,k], score [Q,k]) on every rank.
    The ids travel as int64; scores ride along bit-cast inside the same all-gather."""
    nq = queries.shape[0]
    if dist is None:
        return search_fn(queries, k)
    world, rank = dist.get_world_size(), dist.get_rank()
    b, e, per = shard_bounds(nq, world, rank)
    err = None
    if e > b:
        try:
            idx, sc = search_fn(queries[b:e], k)
        except Exception as ex:  # noqa: BLE001 -- a rank that left here would strand its peers inside the all-gather below: it
            err = ex             # contributes rows marked RANK_FAILED instead and every rank raises once the gather is through
            idx = torch.full((e - b, k), RANK_FAILED, dtype=torch.int64, device=queries.device)
            sc = torch.zeros((e - b, k), dtype=torch.float32, device=queries.device)
    else:
        idx = torch.empty((0, k), dtype=torch.int64, device=queries.device)
        sc = torch.empty((0, k), dtype=torch.float32, device=queries.device)
    packed = torch.cat([idx, sc.contiguous().view(torch.int32).to(torch.int64)], dim=1)  # [q, 2k]
    allp = gather_style_ids(packed, dist, pad_rows=per)
    # drop per-rank padding: rank r contributed rows [r*per, r*per + len_r)
    keep = []
    for r in range(world):
        rb, re, _ = shard_bounds(nq, world, r)
        keep.append(allp[r * per: r * per + (re - rb)])
    if err is not None:
        raise err
    failed = [r for r in range(world) if keep[r].shape[0] and int(keep[r][0, 0]) == RANK_FAILED]
    if failed:
        raise RuntimeError(f"sharded_search: the bank search failed on rank(s) {failed} (their own error is on their stderr)")
    allp = torch.cat(keep, dim=0)
    out_idx = allp[:, :k].contiguous()
    out_sc = allp[:, k:].to(torch.int32).view(torch.float32)
    return out_idx, out_sc


def merge_topk(scores: torch.Tensor, rows: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """k-way merge of per-rank candidate lists: ``scores`` fp64 ``[Q, C]``, ``rows`` int64 ``[Q, C]`` (global row indices,
    -1 = empty slot) -> the best ``k`` per query in the oracle's total order (score descending, row ascending; the order of
    oracle/knn.py::topk_from_scores).  Two stable sorts: by row ascending, then by score descending."""
    big = torch.iinfo(torch.int64).max
    r = torch.where(rows < 0, torch.full_like(rows, big), rows)
    s = torch.where(rows < 0, torch.full_like(scores, float("-inf")), scores)
    o1 = torch.argsort(r, dim=1, stable=True)
    r, s = torch.gather(r, 1, o1), torch.gather(s, 1, o1)
    o2 = torch.argsort(-s, dim=1, stable=True)
    r, s = torch.gather(r, 1, o2)[:, :k], torch.gather(s, 1, o2)[:, :k]
    return torch.where(r == big, torch.full_like(r, -1), r), s


def bank_sharded_search(local_search_fn: Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]], queries: torch.Tensor, k: int,
                        row_offset: int, dist=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Bank-sharded retrieval.  ``local_search_fn(queries, k) -> (idx int64 [Q,k] LOCAL row indices (-1 = none), score fp64
    [Q,k])`` searches this rank's rows (``StyleBank.search_device(q, k, return_f64=True)`` -> ``(idx, _, s64)`` on GPUs);
    ``row_offset`` is the global index of this rank's first row.  Every rank passes the FULL query batch.  Returns the global
    (idx int64 [Q,k], score fp64 [Q,k]) on every rank -- identical to an unsharded search: fp64 scores order candidates of
    different ranks exactly as the oracle does."""
    idx, sc = local_search_fn(queries, k)
    gidx = torch.where(idx >= 0, idx + int(row_offset), idx)
    if dist is None:
        return merge_topk(sc.to(torch.float64), gidx, k)
    world = dist.get_world_size()
    packed = torch.cat([gidx, sc.to(torch.float64).contiguous().view(torch.int64)], dim=1).contiguous()     # [Q, 2k] int64
    allp = torch.empty((world,) + tuple(packed.shape), dtype=torch.int64, device=packed.device)
    dist.all_gather_into_tensor(allp.view(world * packed.shape[0], packed.shape[1]), packed)
    rows = allp[:, :, :k].permute(1, 0, 2).reshape(packed.shape[0], world * k)
    scores = allp[:, :, k:].contiguous().view(torch.float64).permute(1, 0, 2).reshape(packed.shape[0], world * k)
    return merge_topk(scores, rows, k)


# --------------------------------------------------------------------------------------------- driver-side helpers
def init_from_env(backend: Optional[str] = None):
    """The process group of a driver launched as ``python -m torch.distributed.run --nproc-per-node N -m astts.cli.<driver>`` (one
    process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).  -> (dist, rank, world, local_rank) with
    ``dist`` = ``torch.distributed`` once initialised, ``None`` in a plain one-process run (no WORLD_SIZE, or WORLD_SIZE = 1 and no
    ASTTS_FORCE_DIST).  Backend: "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo" (the CPU tests); ``ASTTS_DIST_BACKEND``
    or ``backend`` overrides.  The caller selects its GPU (``cuda:local_rank``) BEFORE the first collective."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("ASTTS_FORCE_DIST") != "1":
        return None, 0, 1, 0
    import torch.distributed as dist

    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world == 1:              # a one-rank group (ASTTS_FORCE_DIST=1): any free port -- two such jobs on one host must not collide
                import socket
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            else:                       # several ranks without a launcher-provided port have to agree on one
                os.environ["MASTER_PORT"] = "29533"
        be = backend or os.environ.get("ASTTS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if be == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(be, rank=rank, world_size=world)
    return dist, rank, world, local


class rank_work:
    """``with parallel.rank_work(dist, "what"):`` around the part of a data-parallel driver that a rank does ALONE (its shard of the
    rows; no collective inside).  At the end of the block the ranks agree on one flag (an all-reduce of one int32 -- it takes the
    place of the closing barrier): a rank whose block raised re-raises its own error, the others raise a RuntimeError naming the
    failed ranks -- nobody waits in a barrier for a peer that has gone (a bad wav path on one rank stalled the
    other seven until the collective timeout).  ``dist = None``: a plain ``with`` block."""

    def __init__(self, dist, what: str = "data-parallel section"):
        self.dist, self.what = dist, what

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if self.dist is None:
            return False
        dist = self.dist
        flag = torch.zeros(dist.get_world_size(), dtype=torch.int32, device=comm_device(dist))
        if et is not None:
            flag[dist.get_rank()] = 1
        dist.all_reduce(flag)
        if et is not None:
            return False                 # this rank's own exception propagates
        failed = [r for r, f in enumerate(flag.tolist()) if f]
        if failed:
            raise RuntimeError(f"{self.what}: rank(s) {failed} of {dist.get_world_size()} failed (their error is on their stderr); "
                               f"rank {dist.get_rank()}'s own rows are complete")
        return False


def broadcast_object(obj, dist, src: int = 0):
    """One small Python object from ``src`` to every rank (e.g. the time stamp of a result directory, so that the ranks agree)."""
    if dist is None:
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def comm_device(dist) -> torch.device:
    """Where a tensor must live to enter a collective of ``dist``'s backend."""
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shutdown() -> None:
    """End of a driver process: leave the process group if one was formed."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
