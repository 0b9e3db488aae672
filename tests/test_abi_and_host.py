"""CPU tests: the C-ABI library loads and exports every symbol include/astts.h declares (no compute
calls without a GPU), and the host-side logic (Milvus-Lite reader, MilvusClient shim surface)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    syms = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            syms |= set(re.findall(r"\b(astts_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    from astts import _lib

    assert os.path.exists(_lib.LIB_PATH), "libastts.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert "astts_knn_search" in syms and len(syms) >= 8
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    # the ctypes table mirrors the header one to one
    import astts.frontend_nets  # noqa: F401
    import astts.ops  # noqa: F401  (register their parts of the ABI)
    assert set(_lib.declared_symbols()) == set(syms)
    lib2 = _lib.load()
    assert lib2.astts_abi_version() == 5


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from astts.knn import StyleBank

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        StyleBank(np.zeros((4, 64), np.float16))


def test_milvus_lite_reader_on_shipped_db(golden_dir, real_bank):
    from astts.milvus_lite import MilvusLiteFile

    f = MilvusLiteFile(os.path.join(golden_dir, "milvus_demo.db"))
    assert sorted(f.collections()) == ["demo_collection", "embeddings_biographies_collection"]
    info = f.info("embeddings_biographies_collection")
    assert info.dim == 6144 and info.metric_type == "COSINE" and info.pk_field == "id"
    assert info.index_params["index_type"] == "AUTOINDEX"
    assert [x.name for x in info.fields][:3] == ["id", "vector", "$meta"]
    v, pks, metas = f.load("embeddings_biographies_collection")
    assert v.dtype == np.float32 and np.array_equal(v, real_bank.astype(np.float32))
    assert pks[0] == 1 and len(pks) == 130
    assert metas[0] == {"file_id": "tonight1_0h2m4dot86s_0h2m7dot13s", "text": "What are you talking about?!"}
    assert f.info("demo_collection").dim == 768
    v2, _, _ = f.load("demo_collection")
    assert v2.shape == (0, 768)
    with pytest.raises(KeyError):
        f.info("nope")
    f.close()


def test_milvus_client_surface_cpu(golden_dir, tmp_path):
    import shutil

    from astts.compat.pymilvus import MilvusClient, MilvusException

    db = str(tmp_path / "milvus_demo.db")        # writes go through to the file: work on a copy
    shutil.copy(os.path.join(golden_dir, "milvus_demo.db"), db)
    c = MilvusClient(db)
    assert c.has_collection(collection_name="embeddings_biographies_collection")
    assert not c.has_collection(collection_name="missing")
    info = c.get_collection_info("embeddings_biographies_collection")
    assert info["num_entities"] == 130 and info["fields"][1]["params"]["dim"] == 6144
    # argument validation happens before any GPU work
    with pytest.raises(MilvusException):
        c.search("missing", data=[[0.0] * 6144], limit=1)
    with pytest.raises(MilvusException):
        c.search("embeddings_biographies_collection", data=[[0.0] * 10], limit=1)
    with pytest.raises(MilvusException):
        c.search("embeddings_biographies_collection", data=[[0.0] * 6144], limit=1, metric_type="L2")
    with pytest.raises(MilvusException):
        c.search("embeddings_biographies_collection", data=[[0.0] * 6144], limit=1, anns_field="emb")
    assert c.search("demo_collection", data=[[0.0] * 768], limit=3) == [[]]  # empty collection
    # bank-construction calls of RAG.py:49-57,541-544
    c.create_collection(collection_name="t", dimension=8)
    r = c.insert(collection_name="t", data=[{"id": 1, "vector": [1.0] * 8, "file_id": "a", "text": "x"}])
    assert r["insert_count"] == 1
    with pytest.raises(MilvusException):
        c.insert(collection_name="t", data=[{"id": 2, "vector": [1.0] * 7}])
    c.drop_collection("t")
    assert not c.has_collection("t")
    c.close()


def test_milvus_lite_writer_reproduces_shipped_blobs(golden_dir):
    """Re-encoding the schema, index and all 130 row blobs of the shipped DB gives the same BYTES."""
    import sqlite3

    from astts import milvus_lite as ml

    con = sqlite3.connect(f"file:{os.path.join(golden_dir, 'milvus_demo.db')}?mode=ro&immutable=1", uri=True)
    con.text_factory = bytes
    n_meta = 0
    for name, mt, blob, sf in con.execute("select collection_name, meta_type, blob_field, string_field from collection_meta"):
        name = name.decode()
        if mt == b"schema":
            _, fields = ml._parse_schema(blob)
            dim = int([f for f in fields if f.data_type == ml.DT_FLOAT_VECTOR][0].params["dim"])
            assert ml.encode_schema(name, dim) == blob and sf == b"id"
        else:
            p = ml._parse_index(blob)
            assert ml.encode_index(int(p["dim"]), p["metric_type"], index_id=ml._first(blob, 2)) == blob and sf == b"vector"
        n_meta += 1
    assert n_meta == 4
    n = 0
    for mid, blob in con.execute('select milvus_id, data from "embeddings_biographies_collection" order by id'):
        r = ml.parse_row(blob)
        assert ml.encode_row(r["id"], r["vector"], r["$meta"], r["RowID"], r["Timestamp"]) == blob
        assert mid.decode() == str(r["id"])
        n += 1
    assert n == 130
    con.close()


def test_milvus_client_persists_bank_like_rag_py(tmp_path, real_bank, golden_dir):
    """RAG.py:49-57,541-544: create_collection + insert into a fresh file; a new client sees the same bank."""
    from astts.compat.pymilvus import MilvusClient
    from astts.milvus_lite import MilvusLiteFile

    meta = json.load(open(os.path.join(golden_dir, "style_bank_meta.json")))
    db = str(tmp_path / "fresh.db")
    c = MilvusClient(db)
    assert not c.has_collection(collection_name="bank")
    c.create_collection(collection_name="bank", dimension=6144)
    c.create_collection(collection_name="scratch", dimension=4, metric_type="IP")
    data = [{"id": i + 1, "vector": real_bank[i].astype(np.float32).tolist(), **meta["rows"][i]} for i in range(20)]
    assert c.insert(collection_name="bank", data=data[:12])["insert_count"] == 12
    assert c.insert(collection_name="bank", data=data[12:])["ids"] == list(range(13, 21))
    c.insert(collection_name="scratch", data={"id": 7, "vector": [1, 2, 3, 4], "tag": "a/b \u00e9"})
    c.drop_collection("scratch")
    c.close()
    f = MilvusLiteFile(db)
    assert f.collections() == ["bank"]
    info = f.info("bank")
    assert info.dim == 6144 and info.metric_type == "COSINE" and info.pk_field == "id"
    v, pks, metas = f.load("bank")
    assert np.array_equal(v, real_bank[:20].astype(np.float32)) and list(pks) == list(range(1, 21))
    assert metas == meta["rows"][:20]
    f.close()
    c2 = MilvusClient(db)
    assert c2.get_collection_info("bank")["num_entities"] == 20
    c2.insert(collection_name="bank", data=[{"id": 21, "vector": real_bank[20].astype(np.float32).tolist(), "text": "x"}])
    c2.close()
    assert MilvusClient(db, persist=False).get_collection_info("bank")["num_entities"] == 21


def test_compat_install_aliases():
    import sys

    from astts import compat

    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "pymilvus" or k.startswith("cosyvoice")}
    try:
        compat.install()
        from pymilvus import MilvusClient  # noqa: F401  (milvus/search_embeddings.py:3)
        from cosyvoice.cli.cosyvoice import CosyVoice  # noqa: F401  (tts_with_rag.py:1)
        from cosyvoice.utils.file_utils import load_wav  # noqa: F401  (tts_with_rag.py:2)
    finally:
        for k in list(sys.modules):
            if k == "pymilvus" or k.startswith("cosyvoice"):
                sys.modules.pop(k)
        sys.modules.update(saved)


def test_milvus_client_explicit_schema_variant(tmp_path):
    """milvus/insert_embeddings.py:52-79,519: FieldSchema / CollectionSchema / DataType, create_collection(schema=),
    create_index(...), auto-generated primary keys, VARCHAR fields returned through output_fields."""
    from astts.compat import pymilvus as pm

    db = str(tmp_path / "explicit.db")
    client = pm.MilvusClient(db)
    fields = [pm.FieldSchema(name="id", dtype=pm.DataType.INT64, is_primary=True, auto_id=True),
              pm.FieldSchema(name="file_id", dtype=pm.DataType.VARCHAR, max_length=500, description="File identifier"),
              pm.FieldSchema(name="vector", dtype=pm.DataType.FLOAT_VECTOR, dim=16, description="Combined embedding vector"),
              pm.FieldSchema(name="text", dtype=pm.DataType.VARCHAR, max_length=1000, description="Conversation text")]
    schema = pm.CollectionSchema(fields=fields, description="Embeddings and biographies collection", metric_type="COSINE")
    client.create_collection(collection_name="c", schema=schema)
    client.create_index(collection_name="c", field_name="vector", index_params={"index_type": "IVF_FLAT", "params": {"nlist": 128}})
    rng = np.random.default_rng(0)
    rows = [{"file_id": f"f{i}", "vector": rng.standard_normal(16).astype(np.float32).tolist(), "text": f"t{i}"} for i in range(5)]
    r = client.insert(collection_name="c", data=rows)
    assert r["insert_count"] == 5 and r["ids"] == [1, 2, 3, 4, 5]            # auto_id
    info = client.describe_collection("c")
    assert info["num_entities"] == 5 and info["fields"][1]["params"]["dim"] == 16 and info["metric_type"] == "COSINE"
    client.close()
    c2 = pm.MilvusClient(db)                                               # persisted: metadata comes back from the file
    c = c2._get("c")
    assert c.pks == [1, 2, 3, 4, 5] and c.metas[2] == {"file_id": "f2", "text": "t2"}
    with pytest.raises(pm.MilvusException):
        pm.CollectionSchema(fields=fields[:2])                             # no vector field
