"""``MilvusClient`` call surface of the reference, served by the HBM-resident StyleBank.

Mirrors exactly the calls the reference makes (paths under /root/reference):
  MilvusClient(db_path)                                     milvus/search_embeddings.py:31
  .has_collection(collection_name=)                         src/search_milvus.py:177
  .get_collection_info(name) / .describe_collection(name)   src/search_milvus.py:183
  .search(collection_name=, data=[vec], anns_field="vector", metric_type="COSINE" | param={...},
          limit=, filter=None, output_fields=[...])          milvus/search_embeddings.py:15-22,
                                                            src/search_milvus.py:140-147,
                                                            milvus/search_json.py:246-252
  .create_collection(collection_name, dimension) / .insert(collection_name, data=[{...}]) /
  .drop_collection(collection_name)                          milvus/RAG.py:49-57,541-544
  FieldSchema / CollectionSchema / DataType, .create_collection(schema=CollectionSchema(...)), .create_index(...)
                                                            milvus/insert_embeddings.py:52-79 (auto_id pk, VARCHAR fields)
Result shape: ``list[Q]`` of ``list[k]`` of ``{'id': pk, 'distance': cosine_similarity,
'entity': {field: value}}`` sorted by similarity descending (``distance`` IS the similarity for
COSINE: output_emb/search_results.json holds 0.81..0.95, larger = closer).  Collections created with ``metric_type`` IP / L2
return the inner product (descending) / the squared Euclidean distance (ascending), as Milvus does.  ``filter``: scalar
expressions over the primary key and the dynamic fields (astts.milvus_filter) become a row mask of the search kernels;
``limit`` up to 1024 (32 hits per selection pass over one scan).

The file behind ``db_path`` is read with astts.milvus_lite (SQLite + protobuf wire format); the
vectors go to HBM once per collection and stay there.  Errors raise ``MilvusException`` -- the
reference wraps every call in try/except and degrades to ``[]`` itself
(milvus/search_embeddings.py:24-27), so the shim does not swallow anything.

Extra keys beyond pymilvus: each hit also carries ``'row'`` (row index in the bank == style id,
because the reference's pk restarts per speaker and is not unique: milvus/RAG.py:507).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ..milvus_lite import MilvusLiteFile, MilvusLiteWriter


class DataType:
    """The pymilvus.DataType members the reference's bank builders use (milvus/insert_embeddings.py:53-56)."""
    BOOL, INT8, INT16, INT32, INT64 = 1, 2, 3, 4, 5
    FLOAT, DOUBLE = 10, 11
    VARCHAR, JSON = 21, 23
    FLOAT_VECTOR = 101


class FieldSchema:
    def __init__(self, name: str, dtype: int, description: str = "", is_primary: bool = False, auto_id: bool = False,
                 dim: Optional[int] = None, max_length: Optional[int] = None, **_kw):
        self.name, self.dtype, self.description = name, dtype, description
        self.is_primary, self.auto_id, self.dim, self.max_length = bool(is_primary), bool(auto_id), dim, max_length


class CollectionSchema:
    """Explicit schema (milvus/insert_embeddings.py:52-63).  ``metric_type`` rides on the schema there (pymilvus ignores
    unknown keyword arguments of CollectionSchema; the collection then gets its metric from the index) -- here it selects
    the collection's metric directly, COSINE when absent."""

    def __init__(self, fields: Sequence[FieldSchema], description: str = "", enable_dynamic_field: bool = False, **kw):
        self.fields, self.description, self.enable_dynamic_field = list(fields), description, enable_dynamic_field
        self.metric_type = kw.get("metric_type")
        pks = [f for f in self.fields if f.is_primary]
        vecs = [f for f in self.fields if f.dtype == DataType.FLOAT_VECTOR]
        if len(pks) != 1 or len(vecs) != 1 or not vecs[0].dim:
            raise MilvusException(1, "schema needs exactly one primary key and one FLOAT_VECTOR field with a dim")
        self.primary_field, self.vector_field = pks[0], vecs[0]
        self.auto_id = pks[0].auto_id
        self.dim = int(vecs[0].dim)


class MilvusException(Exception):
    def __init__(self, code: int = 1, message: str = ""):
        super().__init__(f"<MilvusException: (code={code}, message={message})>")
        self.code = code
        self.message = message


class _Collection:
    def __init__(self, name: str, dim: int, metric: str = "COSINE", pk_field: str = "id",
                 vector_field: str = "vector", index_params: Optional[Dict[str, str]] = None):
        self.name = name
        self.dim = int(dim)
        self.metric = metric.upper()
        self.pk_field = pk_field
        self.vector_field = vector_field
        self.index_params = dict(index_params or {})
        self.auto_id = False
        self.vectors: List[np.ndarray] = []
        self.pks: List[int] = []
        self.metas: List[Dict[str, Any]] = []
        self._bank = None  # astts.knn.StyleBank, built lazily, dropped on insert

    def matrix(self) -> np.ndarray:
        if not self.vectors:
            return np.zeros((0, self.dim), np.float32)
        return np.stack(self.vectors).astype(np.float32, copy=False)

    def bank(self):
        if self._bank is None:
            from ..knn import StyleBank  # needs the GPU + libastts.so; raises otherwise

            m = self.matrix()
            m16 = m.astype(np.float16)
            # upload as fp16 when that is lossless (half the HBM traffic of every scan)
            self._bank = StyleBank(m16 if np.array_equal(m16.astype(np.float32), m) else m, metric=self.metric)
        return self._bank


class MilvusClient:
    def __init__(self, uri: str = "./milvus_demo.db", persist: bool = True, **_kwargs):
        """``uri`` is a Milvus-Lite file path, loaded if it exists.  As with pymilvus, create_collection / insert /
        drop_collection are written through to that file (created on first write) unless ``persist=False``."""
        self.uri = uri
        self._persist = bool(persist) and "://" not in uri
        self._writer: Optional[MilvusLiteWriter] = None
        self._colls: Dict[str, _Collection] = {}
        if os.path.exists(uri):
            f = MilvusLiteFile(uri)
            try:
                for name in f.collections():
                    info = f.info(name)
                    c = _Collection(name, info.dim, info.metric_type, info.pk_field,
                                    info.vector_field.name if info.vector_field else "vector",
                                    info.index_params)
                    v, pks, metas = f.load(name)
                    c.vectors = [v[i] for i in range(v.shape[0])]
                    c.pks = [int(x) for x in pks]
                    c.metas = metas
                    self._colls[name] = c
            finally:
                f.close()

    def _file(self) -> Optional[MilvusLiteWriter]:
        if self._persist and self._writer is None:
            self._writer = MilvusLiteWriter(self.uri)
        return self._writer

    # ------------------------------------------------------------------ collection management
    def has_collection(self, collection_name: str, **_kw) -> bool:
        return collection_name in self._colls

    def list_collections(self, **_kw) -> List[str]:
        return list(self._colls)

    def _get(self, name: str) -> _Collection:
        if name not in self._colls:
            raise MilvusException(100, f"collection not found[collection={name}]")
        return self._colls[name]

    def describe_collection(self, collection_name: str, **_kw) -> Dict[str, Any]:
        c = self._get(collection_name)
        return {
            "collection_name": c.name,
            "auto_id": c.auto_id,
            "num_shards": 0,
            "description": "",
            "fields": [
                {"field_id": 100, "name": c.pk_field, "description": "", "type": 5, "params": {}, "is_primary": True},
                {"field_id": 101, "name": c.vector_field, "description": "", "type": 101, "params": {"dim": c.dim}},
            ],
            "enable_dynamic_field": True,
            "num_entities": len(c.pks),
            "metric_type": c.metric,
        }

    get_collection_info = describe_collection  # src/search_milvus.py:183 uses this older name

    def create_collection(self, collection_name: str, dimension: Optional[int] = None,
                          primary_field_name: str = "id", vector_field_name: str = "vector",
                          metric_type: str = "COSINE", schema=None, index_params=None, **_kw) -> None:
        auto_id = False
        if isinstance(schema, CollectionSchema):      # explicit schema: field names, auto id, metric from the schema object
            dimension = schema.dim
            primary_field_name, vector_field_name = schema.primary_field.name, schema.vector_field.name
            metric_type = schema.metric_type or metric_type
            auto_id = schema.auto_id
        elif schema is not None and dimension is None:
            dimension = getattr(schema, "dim", None) or (schema.get("dim") if isinstance(schema, dict) else None)
        if not dimension:
            raise MilvusException(1, "create_collection needs a dimension")
        if collection_name in self._colls:
            return
        self._colls[collection_name] = _Collection(collection_name, int(dimension), metric_type,
                                                   primary_field_name, vector_field_name)
        self._colls[collection_name].auto_id = auto_id
        if self._file():
            self._file().create_collection(collection_name, int(dimension), metric_type, primary_field_name,
                                           vector_field_name)

    def create_index(self, collection_name: str, field_name: Optional[str] = None, index_params=None, **_kw) -> None:
        """milvus/insert_embeddings.py:75-79.  The search here is exact brute force on the GPU, so an index request only
        records its parameters (and a metric, if it names one, for a collection that has no rows yet)."""
        c = self._get(collection_name)
        params = dict(index_params) if isinstance(index_params, dict) else {}
        mt = params.get("metric_type") or (params.get("params") or {}).get("metric_type")
        if mt and not c.pks:
            c.metric = str(mt).upper()
        c.index_params.update({k: str(v) for k, v in params.items()})

    def drop_collection(self, collection_name: str, **_kw) -> None:
        if self._colls.pop(collection_name, None) is not None and self._file():
            self._file().drop_collection(collection_name)

    def insert(self, collection_name: str, data, **_kw) -> Dict[str, Any]:
        c = self._get(collection_name)
        rows = [data] if isinstance(data, dict) else list(data)
        ids = []
        first_new = len(c.pks)
        for r in rows:      # validate the whole request before any row lands (a bad row rejects the request)
            if np.asarray(r[c.vector_field]).shape != (c.dim,):
                raise MilvusException(1100, f"the dim ({np.asarray(r[c.vector_field]).size}) of field data"
                                            f"({c.vector_field}) is not equal to schema dim ({c.dim})")
        for r in rows:
            vec = np.asarray(r[c.vector_field], dtype=np.float32)
            if vec.shape != (c.dim,):
                raise MilvusException(1100, f"the dim ({vec.size}) of field data({c.vector_field}) is not "
                                            f"equal to schema dim ({c.dim})")
            c.vectors.append(vec)
            pk = (max(c.pks) + 1 if c.pks else 1) if c.auto_id else int(r.get(c.pk_field, len(c.pks)))
            c.pks.append(pk)
            ids.append(pk)
            c.metas.append({k: v for k, v in r.items() if k not in (c.vector_field, c.pk_field)})
        c._bank = None
        if self._file():
            self._file().insert(c.name, ((c.pks[i], c.vectors[i], c.metas[i]) for i in range(first_new, len(c.pks))),
                                c.pk_field, c.vector_field)
        return {"insert_count": len(rows), "ids": ids}

    # ------------------------------------------------------------------ search (the hot path)
    def search(self, collection_name: str, data, filter: Optional[str] = "", limit: int = 10,
               output_fields: Optional[Sequence[str]] = None, search_params: Optional[dict] = None,
               anns_field: Optional[str] = None, metric_type: Optional[str] = None,
               param: Optional[dict] = None, **_kw) -> List[List[Dict[str, Any]]]:
        idx, score = self.search_rows(collection_name, data, filter=filter, limit=limit, search_params=search_params,
                                      anns_field=anns_field, metric_type=metric_type, param=param)
        return self.hits_from_rows(collection_name, idx, score, output_fields)

    def search_rows(self, collection_name: str, data, filter: Optional[str] = "", limit: int = 10, search_params: Optional[dict] = None,
                    anns_field: Optional[str] = None, metric_type: Optional[str] = None, param: Optional[dict] = None, **_kw):
        """The arithmetic half of ``search``: -> (row index int64 [Q, k], cosine similarity fp32 [Q, k]) as numpy arrays, -1 = no hit
        (k = min(limit, rows); k = 0 columns for an empty collection).  Row index == style id.  The data-parallel drivers shard
        THIS call over the ranks and all-gather its result (astts.parallel.sharded_search); ``hits_from_rows`` turns rows into the
        pymilvus result shape wherever the records are written."""
        c = self._get(collection_name)
        mask = None
        if filter:          # (the reference passes None: milvus/search_json.py:246-252)
            from ..milvus_filter import FilterSyntaxError, row_mask
            try:
                mask = row_mask(filter, c.pk_field, c.pks, c.metas)
            except FilterSyntaxError as e:
                raise MilvusException(1100, f"failed to create query plan: cannot parse expression: {filter}, error: {e}") from None
        if anns_field is not None and anns_field != c.vector_field:
            raise MilvusException(1, f"anns_field {anns_field!r} does not exist (vector field is {c.vector_field!r})")
        mt = metric_type or (search_params or {}).get("metric_type") or (param or {}).get("metric_type")
        if mt is not None and mt.upper() != c.metric:
            raise MilvusException(1100, f"metric type not match: invalid parameter[expected={c.metric}][actual={mt}]")
        q = np.asarray(data, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != c.dim:
            raise MilvusException(1100, f"vector dimension mismatch, expected vector size(byte) {c.dim * 4}, "
                                        f"actual {q.shape[-1] * 4}")
        if len(c.pks) == 0:
            return np.zeros((q.shape[0], 0), np.int64), np.zeros((q.shape[0], 0), np.float32)
        if limit < 1:
            raise MilvusException(1, f"limit {limit} is invalid")
        k = min(int(limit), len(c.pks))
        from .._lib import KNN_MAX_K
        if k > KNN_MAX_K:
            # pymilvus accepts limit up to 16384; the reference asks for 1 or 3 (milvus/search_json.py:411,
            # milvus/search_embeddings.py:64).  Here: 32 certified hits per selection pass, at most 32 passes.
            raise MilvusException(1100, f"limit {limit} is not supported by this build: at most {KNN_MAX_K} hits per query "
                                        f"(collection holds {len(c.pks)} rows)")
        idx, score = c.bank().search(q, k, row_mask=mask) if mask is not None else c.bank().search(q, k)
        return np.asarray(idx, dtype=np.int64), np.asarray(score, dtype=np.float32)

    def hits_from_rows(self, collection_name: str, idx, score, output_fields: Optional[Sequence[str]] = None) -> List[List[Dict[str, Any]]]:
        """(row index [Q, k], similarity [Q, k]) -> ``list[Q]`` of ``list[<= k]`` of {'id', 'distance', 'entity', 'row'}."""
        c = self._get(collection_name)
        idx, score = np.asarray(idx), np.asarray(score)
        out: List[List[Dict[str, Any]]] = []
        for qi in range(idx.shape[0]):
            hits = []
            for j in range(idx.shape[1]):
                row = int(idx[qi, j])
                if row < 0:
                    continue
                meta = c.metas[row]
                if output_fields:
                    ent = {f: meta[f] for f in output_fields if f in meta}
                else:
                    ent = {}
                hits.append({"id": c.pks[row], "distance": float(score[qi, j]), "entity": ent, "row": row})
            out.append(hits)
        return out

    def close(self) -> None:
        for c in self._colls.values():
            if c._bank is not None:
                c._bank.close()
                c._bank = None
        if self._writer is not None:
            self._writer.close()
            self._writer = None
