#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's DATA artefacts.

Run in the build container only (needs /root/reference):
    python tests/golden/make_fixtures.py

Inputs (data files of the reference, no source code is copied):
  /root/reference/milvus/milvus_demo.db            the shipped style bank (Milvus-Lite SQLite)
  /root/reference/output_emb/search_results.json   recorded retrieval run = hand-off format of tts_with_rag.py
  /root/reference/data/iemocap.test.json           IEMOCAP test split (text only)

Outputs (tests/golden/):
  style_bank_130x6144.f16.npy   the bank, lossless as fp16 (every fp32 value is fp16-exact)
  style_bank_meta.json          pk ids, {file_id,text} per row, collection/index meta, sha256 of the payload
  knn_kats.json                 known-answer tests: fp64 cosine, (score desc, row asc) order, from oracle/knn.py
  search_results.jsonl          copy of the 64-row retrieval hand-off file (data)
  iemocap_test_sentences.json   the sentences the bench/test configs draw text from
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "autostyle-tts_amd"))

from astts.milvus_lite import MilvusLiteFile  # noqa: E402
from oracle import knn as oknn  # noqa: E402

REF = "/root/reference"
COLL = "embeddings_biographies_collection"

# values recorded by the survey -- the extraction must reproduce them
SHA_F32 = "5c0a7bac58890c26836193749fa68a150a997392a47c2ad896f9db0c8a77b81a"
SHA_F16 = "cc2dc0b8cfc7a386954726c11e0bdee017b8985daba73ec790c76431dc277c74"


def main():
    db = MilvusLiteFile(os.path.join(REF, "milvus", "milvus_demo.db"))
    info = db.info(COLL)
    v, pks, metas = db.load(COLL)
    assert v.shape == (130, 6144), v.shape
    v16 = v.astype(np.float16)
    assert np.array_equal(v16.astype(np.float32), v), "bank is not fp16-exact"
    sha32 = hashlib.sha256(v.astype("<f4").tobytes()).hexdigest()
    sha16 = hashlib.sha256(v16.astype("<f2").tobytes()).hexdigest()
    assert sha32 == SHA_F32, sha32
    assert sha16 == SHA_F16, sha16
    np.save(os.path.join(HERE, "style_bank_130x6144.f16.npy"), v16)

    demo = db.info("demo_collection")
    meta = {
        "collection": COLL,
        "n": int(v.shape[0]),
        "dim": int(v.shape[1]),
        "metric_type": info.metric_type,
        "index_params": info.index_params,
        "fields": [{"id": f.field_id, "name": f.name, "type": f.data_type, "primary": f.is_primary,
                    "dynamic": f.is_dynamic, "params": f.params} for f in info.fields],
        "sha256_f32_le": sha32,
        "sha256_f16_le": sha16,
        "pk": [int(x) for x in pks],
        "rows": metas,
        "other_collections": {"demo_collection": {"dim": demo.dim, "metric_type": demo.metric_type,
                                                  "n": len(list(db.rows("demo_collection")))}},
    }
    with open(os.path.join(HERE, "style_bank_meta.json"), "w", encoding="utf-8") as f:
        json.dump(meta, f, ensure_ascii=False, indent=1)

    # ---- known-answer tests from the fp64 oracle on the real bank
    idx, sc = oknn.knn_search(v16, v, k=5)
    kats = {"self_top5_idx": idx.tolist(), "self_top5_score": sc.tolist()}
    # biography-only ablation queries (first half zeroed; search_json_ab_bio.py:412)
    qb = v.copy()
    qb[:, :3072] = 0
    bi, bs = oknn.knn_search(v16, qb, k=3)
    kats["bio_only_top3_idx"] = bi.tolist()
    kats["bio_only_top3_score"] = bs.tolist()
    # emotion-only ablation queries, L2-normalised (search_json_ab_text.py:412,420)
    qe = v.copy()
    qe[:, 3072:] = 0
    qe /= np.linalg.norm(qe, axis=1, keepdims=True)
    ei, es = oknn.knn_search(v16, qe.astype(np.float32), k=3)
    kats["emo_only_norm_top3_idx"] = ei.tolist()
    kats["emo_only_norm_top3_score"] = es.tolist()
    with open(os.path.join(HERE, "knn_kats.json"), "w") as f:
        json.dump(kats, f)

    # ---- hand-off file (data) and text source
    rows = []
    with open(os.path.join(REF, "output_emb", "search_results.json"), encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    assert len(rows) == 64
    with open(os.path.join(HERE, "search_results.jsonl"), "w", encoding="utf-8") as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")

    test = json.load(open(os.path.join(REF, "data", "iemocap.test.json"), encoding="utf-8"))
    sents = {"Ses05M_impro03": test["Ses05M_impro03"]["sentences"],
             "all": [s for conv in test.values() for s in conv["sentences"]]}
    assert len(sents["all"]) == 1623, len(sents["all"])
    with open(os.path.join(HERE, "iemocap_test_sentences.json"), "w", encoding="utf-8") as f:
        json.dump(sents, f, ensure_ascii=False)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
