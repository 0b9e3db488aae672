"""Reader and writer for the Milvus-Lite (2.4-era) on-disk format: one SQLite file, one table per
collection, one protobuf ``InsertRequest``-style blob per row.

This is the storage format behind ``MilvusClient("milvus_demo.db")`` in the reference
(/root/reference/milvus/search_embeddings.py:31, /root/reference/src/search_milvus.py:166,
bank built by /root/reference/milvus/RAG.py:49-57,541-544).  Only stdlib is used
(sqlite3 + a protobuf *wire* decoder) -- pymilvus / milvus-lite are not dependencies.

Layout (reverse-engineered from the shipped /root/reference/milvus/milvus_demo.db):
  table ``collection_meta(id, collection_name, meta_type, blob_field, string_field)``
      meta_type 'schema': CollectionSchema {1:name, 4:repeated FieldSchema{1:fieldID, 2:name,
                          3:is_primary, 5:data_type, 6:type_params{1:key,2:value}, 12:is_dynamic}}
      meta_type 'index' : IndexInfo {3:field_name, 5:repeated KeyValuePair{1:key,2:value}}
  table ``<collection>(id INTEGER PK, milvus_id VARCHAR, data BLOB)``
      data = {1: repeated FieldData{1:type, 2:field_name, 3:scalars | 4:vectors, 5:field_id}, 2:num_rows}
      Int64   (type 5)  : 3:scalars{3:long_data{1: packed varint}}
      JSON    (type 23) : 3:scalars{9:json_data{1: bytes}}
      FloatVec(type 101): 4:vectors{1:dim, 2:float_vector{1: packed <f4}}
  The blobs are stored with TEXT affinity: the connection needs ``text_factory = bytes``.
"""
from __future__ import annotations

import json
import sqlite3
from dataclasses import dataclass, field
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

DT_INT64 = 5
DT_VARCHAR = 21
DT_JSON = 23
DT_FLOAT_VECTOR = 101


# ----------------------------------------------------------------------------- protobuf wire
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def iter_fields(buf: bytes) -> Iterator[Tuple[int, int, Any]]:
    """Yield (field_number, wire_type, value) for one protobuf message."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            if len(v) != ln:
                raise ValueError("truncated length-delimited field")
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _first(buf: bytes, fno: int, default=None):
    for f, _, v in iter_fields(buf):
        if f == fno:
            return v
    return default


def _packed_varints(buf: bytes) -> List[int]:
    out, pos = [], 0
    while pos < len(buf):
        v, pos = _varint(buf, pos)
        if v >= 1 << 63:
            v -= 1 << 64
        out.append(v)
    return out


# ----------------------------------------------------------------------------- schema / rows
@dataclass
class FieldInfo:
    field_id: int
    name: str
    data_type: int
    is_primary: bool = False
    is_dynamic: bool = False
    params: Dict[str, str] = field(default_factory=dict)


@dataclass
class CollectionInfo:
    name: str
    fields: List[FieldInfo]
    index_params: Dict[str, str]
    pk_field: str = "id"

    @property
    def vector_field(self) -> Optional[FieldInfo]:
        for f in self.fields:
            if f.data_type == DT_FLOAT_VECTOR:
                return f
        return None

    @property
    def dim(self) -> int:
        vf = self.vector_field
        return int(vf.params.get("dim", 0)) if vf else 0

    @property
    def metric_type(self) -> str:
        return self.index_params.get("metric_type", "COSINE")


def _parse_schema(blob: bytes) -> Tuple[str, List[FieldInfo]]:
    name = ""
    fields: List[FieldInfo] = []
    for fno, wt, v in iter_fields(blob):
        if fno == 1 and wt == 2:
            name = v.decode("utf-8")
        elif fno == 4 and wt == 2:
            fi = FieldInfo(0, "", 0)
            for f2, w2, v2 in iter_fields(v):
                if f2 == 1 and w2 == 0:
                    fi.field_id = v2
                elif f2 == 2 and w2 == 2:
                    fi.name = v2.decode("utf-8")
                elif f2 == 3 and w2 == 0:
                    fi.is_primary = bool(v2)
                elif f2 == 5 and w2 == 0:
                    fi.data_type = v2
                elif f2 == 6 and w2 == 2:
                    k = _first(v2, 1, b"").decode("utf-8")
                    fi.params[k] = _first(v2, 2, b"").decode("utf-8")
                elif f2 == 12 and w2 == 0:
                    fi.is_dynamic = bool(v2)
            fields.append(fi)
    return name, fields


def _parse_index(blob: bytes) -> Dict[str, str]:
    out: Dict[str, str] = {}
    for fno, wt, v in iter_fields(blob):
        if fno == 5 and wt == 2:
            k = _first(v, 1, b"").decode("utf-8")
            out[k] = _first(v, 2, b"").decode("utf-8")
        elif fno == 3 and wt == 2:
            out["field_name"] = v.decode("utf-8")
    return out


def parse_row(blob: bytes) -> Dict[str, Any]:
    """Decode one row blob -> {field_name: value}.  Vectors come back as fp32 ndarrays,
    the dynamic ``$meta`` JSON as a dict."""
    row: Dict[str, Any] = {}
    for fno, wt, v in iter_fields(blob):
        if fno != 1 or wt != 2:
            continue
        ftype = 0
        fname = ""
        scalars = vectors = None
        for f2, w2, v2 in iter_fields(v):
            if f2 == 1 and w2 == 0:
                ftype = v2
            elif f2 == 2 and w2 == 2:
                fname = v2.decode("utf-8")
            elif f2 == 3 and w2 == 2:
                scalars = v2
            elif f2 == 4 and w2 == 2:
                vectors = v2
        if ftype == DT_FLOAT_VECTOR and vectors is not None:
            dim = _first(vectors, 1, 0)
            fv = _first(vectors, 2, b"")
            data = _first(fv, 1, b"")
            arr = np.frombuffer(data, dtype="<f4")
            if dim and arr.size != dim:
                raise ValueError(f"vector field {fname}: {arr.size} values, dim {dim}")
            row[fname] = arr
        elif ftype == DT_INT64 and scalars is not None:
            ld = _first(scalars, 3, b"")
            vals = _packed_varints(_first(ld, 1, b""))
            row[fname] = vals[0] if vals else None
        elif ftype == DT_JSON and scalars is not None:
            jd = _first(scalars, 9, b"")
            payload = _first(jd, 1, b"")
            row[fname] = json.loads(payload.decode("utf-8")) if payload else {}
        elif ftype == DT_VARCHAR and scalars is not None:
            sd = _first(scalars, 6, b"")
            s = _first(sd, 1, b"")
            row[fname] = s.decode("utf-8")
    return row


class MilvusLiteFile:
    """Read-only view of a Milvus-Lite SQLite file."""

    def __init__(self, path: str):
        self.path = path
        self._con = sqlite3.connect(f"file:{path}?mode=ro&immutable=1", uri=True)
        self._con.text_factory = bytes

    def close(self) -> None:
        self._con.close()

    def collections(self) -> List[str]:
        cur = self._con.execute("select distinct collection_name from collection_meta")
        return [r[0].decode("utf-8") for r in cur.fetchall()]

    def info(self, name: str) -> CollectionInfo:
        cur = self._con.execute(
            "select meta_type, blob_field, string_field from collection_meta where collection_name = ?",
            (name,))
        fields: List[FieldInfo] = []
        index: Dict[str, str] = {}
        pk = "id"
        found = False
        for meta_type, blob, sfield in cur.fetchall():
            found = True
            if meta_type == b"schema":
                _, fields = _parse_schema(blob)
                if sfield:
                    pk = sfield.decode("utf-8")
            elif meta_type == b"index":
                index.update(_parse_index(blob))
        if not found:
            raise KeyError(f"collection {name!r} not found in {self.path}")
        return CollectionInfo(name, fields, index, pk)

    def rows(self, name: str) -> Iterator[Dict[str, Any]]:
        if name not in self.collections():
            raise KeyError(f"collection {name!r} not found in {self.path}")
        cur = self._con.execute(f'select id, data from "{name}" order by id')
        for _rowid, blob in cur:
            yield parse_row(bytes(blob))

    def load(self, name: str):
        """-> (vectors fp32 [N,D], pks int64 [N], metas list[dict]) in storage (row) order."""
        info = self.info(name)
        vname = info.vector_field.name if info.vector_field else "vector"
        vecs, pks, metas = [], [], []
        for row in self.rows(name):
            vecs.append(row[vname])
            pks.append(row.get(info.pk_field, len(pks)))
            meta = dict(row.get("$meta", {}) or {})
            for k, v in row.items():
                if k not in (vname, info.pk_field, "$meta", "RowID", "Timestamp"):
                    meta[k] = v
            metas.append(meta)
        d = info.dim
        v = np.stack(vecs).astype(np.float32) if vecs else np.zeros((0, d), np.float32)
        return v, np.asarray(pks, dtype=np.int64), metas


# ----------------------------------------------------------------------------- writer
# The inverse of the reader above, so that banks built through ``MilvusClient.create_collection`` / ``insert``
# (the reference's bank builder, /root/reference/milvus/RAG.py:49-57,541-544) persist in the same file
# format the shipped milvus_demo.db uses.  tests/test_oracle_knn.py re-encodes every schema / index / row
# blob of that file and requires the bytes to be identical.
def _enc_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_key(fno: int, wt: int) -> bytes:
    return _enc_varint((fno << 3) | wt)


def _enc_bytes(fno: int, payload: bytes) -> bytes:
    return _enc_key(fno, 2) + _enc_varint(len(payload)) + payload


def _enc_uint(fno: int, v: int) -> bytes:
    return _enc_key(fno, 0) + _enc_varint(v)


def _enc_kv(fno: int, k: str, v: str) -> bytes:
    return _enc_bytes(fno, _enc_bytes(1, k.encode("utf-8")) + _enc_bytes(2, v.encode("utf-8")))


def encode_schema(name: str, dim: int, pk_field: str = "id", vector_field: str = "vector") -> bytes:
    """CollectionSchema of a quick-setup collection (int64 pk, float vector, dynamic ``$meta``, plus the two
    system fields RowID / Timestamp), proto3 encoding (zero-valued scalars are omitted)."""
    def fschema(fid, fname, dtype, primary=False, desc="", params=(), dynamic=False):
        b = b""
        if fid:
            b += _enc_uint(1, fid)
        b += _enc_bytes(2, fname.encode("utf-8"))
        if primary:
            b += _enc_uint(3, 1)
        if desc:
            b += _enc_bytes(4, desc.encode("utf-8"))
        b += _enc_uint(5, dtype)
        for k, v in params:
            b += _enc_kv(6, k, v)
        if dynamic:
            b += _enc_uint(12, 1)
        return _enc_bytes(4, b)

    out = _enc_bytes(1, name.encode("utf-8"))
    out += fschema(100, pk_field, DT_INT64, primary=True)
    out += fschema(101, vector_field, DT_FLOAT_VECTOR, params=(("dim", str(int(dim))),))
    out += fschema(102, "$meta", DT_JSON, desc="dynamic schema", dynamic=True)
    out += fschema(0, "RowID", DT_INT64, desc="row id")
    out += fschema(1, "Timestamp", DT_INT64, desc="time stamp")
    out += _enc_uint(5, 1)          # enable_dynamic_field
    return out


def encode_index(dim: int, metric_type: str = "COSINE", vector_field: str = "vector", index_id: int = 0,
                 index_type: str = "AUTOINDEX", m: int = 18, ef_construction: int = 240) -> bytes:
    out = _enc_uint(1, 101) + _enc_uint(2, index_id) + _enc_bytes(3, vector_field.encode("utf-8"))
    out += _enc_kv(5, "M", str(m)) + _enc_kv(5, "efConstruction", str(ef_construction))
    out += _enc_kv(5, "index_type", index_type) + _enc_kv(5, "metric_type", metric_type)
    out += _enc_kv(5, "dim", str(int(dim)))
    out += _enc_uint(6, 1)
    return out


def encode_row(pk: int, vector: np.ndarray, meta: Dict[str, Any], row_id: int, timestamp: int,
               pk_field: str = "id", vector_field: str = "vector") -> bytes:
    """One single-row InsertRequest-style blob (see the module docstring for the layout)."""
    def longs(fid, fname, val):
        sc = _enc_bytes(3, _enc_bytes(1, _enc_varint(int(val))))
        b = _enc_uint(1, DT_INT64) + _enc_bytes(2, fname.encode("utf-8")) + _enc_bytes(3, sc)
        if fid:
            b += _enc_uint(5, fid)
        return _enc_bytes(1, b)

    vec = np.ascontiguousarray(vector, dtype="<f4").reshape(-1)
    vf = _enc_uint(1, vec.size) + _enc_bytes(2, _enc_bytes(1, vec.tobytes()))
    vb = _enc_uint(1, DT_FLOAT_VECTOR) + _enc_bytes(2, vector_field.encode("utf-8")) + _enc_bytes(4, vf) + _enc_uint(5, 101)
    # ujson conventions, as the shipped file shows them: compact separators, raw UTF-8, "/" written "\\/"
    payload = json.dumps(meta, separators=(",", ":"), ensure_ascii=False).replace("/", "\\/").encode("utf-8")
    jb = _enc_uint(1, DT_JSON) + _enc_bytes(2, b"$meta") + _enc_bytes(3, _enc_bytes(9, _enc_bytes(1, payload))) + _enc_uint(5, 102)
    out = longs(100, pk_field, pk) + _enc_bytes(1, vb) + _enc_bytes(1, jb)
    out += longs(0, "RowID", row_id) + longs(1, "Timestamp", timestamp)
    out += _enc_uint(2, 1)          # num_rows
    return out


class MilvusLiteWriter:
    """Creates / appends to a Milvus-Lite SQLite file (same tables and blobs as the shipped milvus_demo.db)."""

    def __init__(self, path: str):
        self.path = path
        self._con = sqlite3.connect(path)
        self._con.execute("CREATE TABLE IF NOT EXISTS collection_meta (id INTEGER PRIMARY KEY, collection_name "
                          "VARCHAR(1024), meta_type VARCHAR(1024), blob_field BLOB, string_field VARCHAR(1024))")
        self._con.commit()
        self._tick = 0

    def close(self) -> None:
        self._con.commit()
        self._con.close()

    def _hybrid_ts(self) -> int:
        import time
        self._tick += 1
        return (int(time.time() * 1000) << 18) + (self._tick & 0x3FFFF)       # TSO layout: physical ms << 18 | logical

    def has_collection(self, name: str) -> bool:
        cur = self._con.execute("select 1 from collection_meta where collection_name = ? limit 1", (name,))
        return cur.fetchone() is not None

    def create_collection(self, name: str, dim: int, metric_type: str = "COSINE", pk_field: str = "id",
                          vector_field: str = "vector") -> None:
        if '"' in name:
            raise ValueError("collection name must not contain quotes")
        if self.has_collection(name):
            return
        import random
        ins = "insert into collection_meta (collection_name, meta_type, blob_field, string_field) values (?,?,?,?)"
        self._con.execute(ins, (name, "schema", encode_schema(name, dim, pk_field, vector_field), pk_field))
        self._con.execute(ins, (name, "index", encode_index(dim, metric_type.upper(), vector_field,
                                                            random.getrandbits(62)), vector_field))
        self._con.execute(f'CREATE TABLE IF NOT EXISTS "{name}" (id INTEGER PRIMARY KEY, milvus_id VARCHAR(1024), data BLOB)')
        self._con.commit()

    def drop_collection(self, name: str) -> None:
        if '"' in name:
            raise ValueError("collection name must not contain quotes")
        self._con.execute("delete from collection_meta where collection_name = ?", (name,))
        self._con.execute(f'DROP TABLE IF EXISTS "{name}"')
        self._con.commit()

    def insert(self, name: str, rows, pk_field: str = "id", vector_field: str = "vector") -> None:
        """``rows``: iterable of (pk, vector, meta dict)."""
        recs = []
        for pk, vec, meta in rows:
            ts = self._hybrid_ts()
            recs.append((str(int(pk)), encode_row(pk, vec, meta, ts, ts, pk_field, vector_field)))
        self._con.executemany(f'insert into "{name}" (milvus_id, data) values (?,?)', recs)
        self._con.commit()
