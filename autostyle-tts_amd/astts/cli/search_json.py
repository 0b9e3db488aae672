"""Batch retrieval -> JSONL hand-off file: the search + record-writing half of /root/reference/milvus/search_json.py
(:382-461).  The LLM half of that script (biography / emotion-label generation + Llama embedding, :313-381) is outside
the hot path: this driver takes the 6144-d query vectors precomputed, one per input row.

    python -m astts.cli.search_json --input_json in.jsonl --query_npy q.npy --db_path milvus_demo.db \\
        --output_file search_results.json [--file_prefix_path /data/seg_wav]

Input rows {zh_text, speaker}; output rows {zh_text, speaker, retrieved_file_id, retrieved_text, distance}
(top-1, ``distance`` = cosine similarity), "N/A" rows when nothing is found, "Error" rows on failure -- as :423-449.
All queries of the file go to the GPU as ONE batch (the reference loops them one by one).

Data-parallel form (BASELINE config 4: the IEMOCAP test set over the 8 GPUs of a node):

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m astts.cli.search_json --input_json ... (same flags)

one process per GPU, the bank replicated; rank r searches rows [r ceil(Q/W), (r+1) ceil(Q/W)) of the input, ONE all-gather of the
retrieved (style id, similarity) pairs over RCCL (astts.parallel.sharded_search), and rank 0 writes the JSONL in input order -- the
file is identical to the one-process run's.
"""
import argparse
import json
import os
import traceback

import numpy as np

from astts import parallel
from astts.compat.pymilvus import MilvusClient


def read_input_json(path):
    rows = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    return rows


def main(args, client=None):
    import torch

    dist, rank, world, local = parallel.init_from_env()
    if dist is not None and dist.get_backend() == "nccl":
        torch.cuda.set_device(local)
    client = client or MilvusClient(args.db_path)
    rows = read_input_json(args.input_json)
    q = np.load(args.query_npy).astype(np.float32)
    if q.shape[0] != len(rows):
        raise SystemExit(f"{args.query_npy}: {q.shape[0]} vectors for {len(rows)} input rows")
    # rows without text are skipped, as the reference does (milvus/search_json.py:385-387) -- together with their query vector
    keep = [i for i, r in enumerate(rows) if r.get("zh_text", "").strip()]
    if rank == 0:
        for i in sorted(set(range(len(rows))) - set(keep)):
            print(f"Skipping empty text for speaker '{rows[i].get('speaker', 'UNKNOWN_SPEAKER')}'.")
    rows = [rows[i] for i in keep]
    q = q[keep]
    results = []
    hits = []
    if len(rows):
        dev = parallel.comm_device(dist)

        def search_fn(q_local, k):      # this rank's shard of the queries against its replica of the bank
            idx, score = client.search_rows(collection_name=args.collection_name, data=q_local.cpu().numpy(), limit=k, filter=None)
            if idx.shape[1] < k:        # fewer rows than k (an empty collection): pad, -1 = no hit
                pad = k - idx.shape[1]
                idx = np.concatenate([idx, np.full((idx.shape[0], pad), -1, np.int64)], 1)
                score = np.concatenate([score, np.zeros((score.shape[0], pad), np.float32)], 1)
            return torch.from_numpy(idx).to(dev), torch.from_numpy(score).to(dev)

        try:
            idx, score = parallel.sharded_search(search_fn, torch.from_numpy(q).to(dev), 1, dist)
            hits = client.hits_from_rows(args.collection_name, idx.cpu().numpy(), score.cpu().numpy(), ["file_id", "text"])
        except Exception as e:  # noqa: BLE001
            if dist is not None:        # a rank that fails alone would leave the others inside the collective: fail the job
                raise
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
    if args.output_file and rank == 0:      # every rank holds every record after the all-gather; one of them writes
        out_dir = os.path.dirname(args.output_file)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        with open(args.output_file, "w", encoding="utf-8") as f:
            for r in results:
                f.write(json.dumps(r, ensure_ascii=False) + "\n")
        print(f"Search results saved to '{args.output_file}'.")
    if dist is not None:
        dist.barrier()
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
    parallel.shutdown()
