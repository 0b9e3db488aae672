"""Style bank resident in HBM + brute-force kNN -- COSINE (the reference's collection), IP, L2 -- (host side of astts_knn_*).

This is the engine under the ``MilvusClient.search`` shim (astts/compat/pymilvus.py).  It
replaces what the reference gets from milvus-lite for
/root/reference/milvus/search_embeddings.py:15-22 and /root/reference/src/search_milvus.py:140-147.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib


def _require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("astts.knn needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback in the product path")


class StyleBank:
    """``N x D`` style-embedding bank held on one GPU.

    ``vectors``: numpy / torch array ``[N, D]`` (fp16 or fp32).  Row index == style id.
    """

    def __init__(self, vectors, device: Optional[torch.device] = None, metric: str = "COSINE"):
        _require_gpu()
        lib = _lib.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if isinstance(vectors, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(vectors))
        else:
            t = vectors
        if t.dim() != 2:
            raise ValueError(f"bank must be [N, D], got {tuple(t.shape)}")
        if t.dtype not in (torch.float16, torch.float32):
            t = t.to(torch.float32)
        metric_id = {"COSINE": _lib.METRIC_COSINE, "IP": _lib.METRIC_IP, "L2": _lib.METRIC_L2}.get(metric.upper())
        if metric_id is None:
            raise ValueError(f"unknown metric {metric!r}")
        with torch.cuda.device(self.device):
            src = t.to(self.device).contiguous()
            h = ctypes.c_void_p()
            _lib.check(lib.astts_knn_create(
                src.data_ptr(), src.shape[0], src.shape[1],
                _lib.DTYPE_F16 if src.dtype == torch.float16 else _lib.DTYPE_F32,
                metric_id, _lib.stream_ptr(), ctypes.byref(h)))
        self._h = h
        self.metric = metric.upper()
        self.n, self.d = int(src.shape[0]), int(src.shape[1])
        exact = ctypes.c_int32()
        _lib.check(lib.astts_knn_info(self._h, None, None, ctypes.byref(exact)))
        self.scan_plane_exact = bool(exact.value)
        self._ws: Optional[torch.Tensor] = None
        self._ws_key: Tuple[int, int] = (0, 0)
        self._plans: dict = {}                       # (nq, k) -> (workspace tensor, aligned pointer, bytes)
        self._search = lib.astts_knn_search_masked

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().astts_knn_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def _workspace(self, nq: int, k: int) -> torch.Tensor:
        need = int(_lib.load().astts_knn_workspace_bytes(self._h, nq, k))
        if need == 0:
            raise ValueError(f"invalid search shape nq={nq} k={k}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._plans = {}                         # (plans of the smaller workspace point into freed memory)
        return self._ws

    def search_device(self, queries: torch.Tensor, k: int, force_exact: bool = False,
                      out_idx: Optional[torch.Tensor] = None, out_score: Optional[torch.Tensor] = None, return_f64: bool = False,
                      row_mask: Optional[torch.Tensor] = None):
        """queries: fp32 ``[Q, D]`` on this bank's GPU.  Returns (idx int64 [Q,k], score fp32 [Q,k])
        on the GPU, enqueued on the current stream (no synchronisation), closest first (COSINE / IP: score descending; L2:
        squared distance ascending; ties by row index).  ``return_f64``: a third tensor with the fp64 scores (what a
        bank-sharded search merges on: astts.parallel.bank_sharded_search).  ``row_mask``: uint8 / bool ``[N]`` (one mask for
        every query) or ``[Q, N]`` on the GPU -- only rows with a non-zero entry can be hits (a Milvus ``filter``)."""
        if queries.dim() != 2 or queries.shape[1] != self.d:
            raise ValueError(f"queries must be [Q, {self.d}], got {tuple(queries.shape)}")
        if not 1 <= k <= _lib.KNN_MAX_K:
            raise ValueError(f"k must be in 1..{_lib.KNN_MAX_K}, got {k}")
        # (a config-2 search is ~12 us of GPU time: the host side of this call is kept to a handful of attribute reads and ONE ctypes
        # call -- the workspace and its size are cached per (queries, k), tensors that already are what the ABI takes are passed as is)
        q = queries
        if q.dtype != torch.float32 or q.device != self.device or not q.is_contiguous():
            q = queries.to(device=self.device, dtype=torch.float32).contiguous()
        nq = int(q.shape[0])
        if nq == 0:
            e = (torch.empty((0, k), dtype=torch.int64, device=self.device), torch.empty((0, k), dtype=torch.float32, device=self.device))
            return e + (torch.empty((0, k), dtype=torch.float64, device=self.device),) if return_f64 else e
        switch = torch.cuda.current_device() != self.device.index
        if switch:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(self.device)
        try:
            plan = self._plans.get((nq, k))
            if plan is None:
                ws = self._workspace(nq, k)
                base = ws.data_ptr()
                aligned = (base + 255) // 256 * 256
                plan = self._plans[(nq, k)] = (ws, aligned, ws.numel() - (aligned - base))
            ws, aligned, ws_bytes = plan
            if out_idx is None:
                out_idx = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            if out_score is None:
                out_score = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            s64 = torch.empty((nq, k), dtype=torch.float64, device=self.device) if return_f64 else None
            mptr, mstride = None, 0
            if row_mask is not None:
                m = row_mask.to(device=self.device)
                m = (m if m.dtype == torch.uint8 else (m != 0).to(torch.uint8)).contiguous()
                if m.shape == (self.n,):
                    mstride = 0
                elif m.shape == (nq, self.n):
                    mstride = self.n
                else:
                    raise ValueError(f"row_mask must be [{self.n}] or [{nq}, {self.n}], got {tuple(m.shape)}")
                mptr = m.data_ptr()
            rc = self._search(self._h, q.data_ptr(), nq, k, out_idx.data_ptr(), out_score.data_ptr(), None if s64 is None else s64.data_ptr(),
                              mptr, mstride, aligned, ws_bytes, _lib.KNN_FORCE_EXACT if force_exact else 0,
                              torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                _lib.check(rc)
        finally:
            if switch:
                torch.cuda.set_device(prev)
        return (out_idx, out_score, s64) if return_f64 else (out_idx, out_score)

    def search(self, queries, k: int, force_exact: bool = False, row_mask=None):
        """Host convenience: accepts numpy / lists, returns numpy (idx int64 [Q,k], score fp32 [Q,k])."""
        q = torch.as_tensor(np.asarray(queries, dtype=np.float32))
        if q.dim() == 1:
            q = q[None, :]
        if row_mask is not None and not isinstance(row_mask, torch.Tensor):
            row_mask = torch.from_numpy(np.ascontiguousarray(np.asarray(row_mask) != 0).astype(np.uint8))
        idx, sc = self.search_device(q.to(self.device), k, force_exact, row_mask=row_mask)
        return idx.cpu().numpy(), sc.cpu().numpy()

    def last_fallbacks(self) -> int:
        """How many queries of the last search needed the exact fp64 scan (synchronises)."""
        if self._ws is None:
            return 0
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        out = ctypes.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().astts_knn_last_fallbacks(self._h, aligned, _lib.stream_ptr(), ctypes.byref(out)))
        return int(out.value)

    # ---- bench-only: time the scan kernel with HIP events on the search stream
    def profile_enable(self, on: bool = True) -> None:
        _lib.check(_lib.load().astts_knn_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms = ctypes.c_double()
        cnt = ctypes.c_int64()
        _lib.check(_lib.load().astts_knn_profile_read(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return float(ms.value), int(cnt.value)
