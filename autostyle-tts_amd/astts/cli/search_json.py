"""Batch retrieval -> JSONL hand-off file: the search + record-writing half of /root/reference/milvus/search_json.py
(:382-461).  The LLM half of that script (biography / emotion-label generation + Llama embedding, :313-381) is outside
the hot path: this driver takes the 6144-d query vectors precomputed, one per input row.

    python -m astts.cli.search_json --input_json in.jsonl --query_npy q.npy --db_path milvus_demo.db \\
        --output_file search_results.json [--file_prefix_path /data/seg_wav]

Input rows {zh_text, speaker}; output rows {zh_text, speaker, retrieved_file_id, retrieved_text, distance}
(top-1, ``distance`` = cosine similarity), "N/A" rows when nothing is found, "Error" rows on failure -- as :423-449.
All queries of the file go to the GPU as ONE batch (the reference loops them one by one).
"""
import argparse
import json
import os
import traceback

import numpy as np

from astts.compat.pymilvus import MilvusClient


def read_input_json(path):
    rows = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    return rows


def main(args):
    client = MilvusClient(args.db_path)
    rows = read_input_json(args.input_json)
    q = np.load(args.query_npy).astype(np.float32)
    if q.shape[0] != len(rows):
        raise SystemExit(f"{args.query_npy}: {q.shape[0]} vectors for {len(rows)} input rows")
    # rows without text are skipped, as the reference does (milvus/search_json.py:385-387) -- together with their query vector
    keep = [i for i, r in enumerate(rows) if r.get("zh_text", "").strip()]
    for i in sorted(set(range(len(rows))) - set(keep)):
        print(f"Skipping empty text for speaker '{rows[i].get('speaker', 'UNKNOWN_SPEAKER')}'.")
    rows = [rows[i] for i in keep]
    q = q[keep]
    results = []
    try:
        hits = client.search(collection_name=args.collection_name, data=q, limit=1, filter=None,
                             output_fields=["file_id", "text"]) if len(rows) else []
    except Exception as e:  # noqa: BLE001
        print(f"Error during search: {e}")
        traceback.print_exc()
        hits = None
    for i, sample in enumerate(rows):
        zh_text = sample.get("zh_text", "").strip()
        speaker = sample.get("speaker", "UNKNOWN_SPEAKER")
        if hits is None:
            rec = {"zh_text": zh_text, "speaker": speaker, "retrieved_file_id": "Error", "retrieved_text": "Error", "distance": "Error"}
        elif hits[i]:
            top = hits[i][0]
            fid = top.get("entity", {}).get("file_id", "N/A")
            rec = {"zh_text": zh_text, "speaker": speaker,
                   "retrieved_file_id": os.path.join(args.file_prefix_path, fid) if args.file_prefix_path else fid,
                   "retrieved_text": top.get("entity", {}).get("text", "N/A"), "distance": top.get("distance", "N/A")}
        else:
            rec = {"zh_text": zh_text, "speaker": speaker, "retrieved_file_id": "N/A", "retrieved_text": "N/A", "distance": "N/A"}
        results.append(rec)
    if args.output_file:
        out_dir = os.path.dirname(args.output_file)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        with open(args.output_file, "w", encoding="utf-8") as f:
            for r in results:
                f.write(json.dumps(r, ensure_ascii=False) + "\n")
        print(f"Search results saved to '{args.output_file}'.")
    return results


def build_parser():
    p = argparse.ArgumentParser(description="Batch style retrieval -> JSONL hand-off")
    p.add_argument("--input_json", required=True)
    p.add_argument("--query_npy", required=True, help="[rows, dim] float32 query vectors (emotion | biography halves)")
    p.add_argument("--db_path", default="milvus_demo.db")
    p.add_argument("--collection_name", default="embeddings_biographies_collection")
    p.add_argument("--output_file", default="")
    p.add_argument("--file_prefix_path", default="")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
