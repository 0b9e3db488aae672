"""Batch retrieval -> JSONL hand-off file: /root/reference/milvus/search_json.py (:313-461) on one GPU in one command.

    python -m astts.cli.search_json --input_json in.jsonl --model_path /path/to/llama-3.2-3b [--biography_json bios.json] \
        --db_path milvus_demo.db --output_file search_results.json [--file_prefix_path /data/seg_wav]

Per input row {zh_text, speaker} (:382-461): emotion label (greedy continuation of the few-shot prompt, :154-198) -> combined
embedding [label | speaker biography] through the Llama embedder (mean-pooled last hidden state, :76-109, :201-229) -> top-1 COSINE
search -> {zh_text, speaker, retrieved_file_id, retrieved_text, distance} (``distance`` = cosine similarity), "N/A" rows when nothing
is found, "Error" rows on failure -- as :423-449.  Text -> 6144-d -> style id never leaves the GPU box: the labels of a batch of rows
come from ONE KV-cached greedy decode (``LlamaEmbedder.generate_emotion_labels``), the distinct label and biography texts are
embedded once each in padded batches, and all queries of the file go to the retrieval kernel as ONE batch (the reference loops
rows one by one, re-embedding the same six labels and the same biography every time).

Speaker biographies: the reference SAMPLES them from the LLM (:113-151, do_sample=True, 250 tokens: not reproducible, and outside
the hot path -- SURVEY.md 2 #9); here they come from ``--biography_json`` ({speaker: biography}, e.g. the output of the reference's
own bank construction) and fall back to the reference's own fallback text "This is a placeholder biography." (:375, :400).

``--query_npy q.npy`` ([rows, dim] float32) replaces the LLM half with precomputed query vectors, one per input row.

Data-parallel form (BASELINE config 4: the IEMOCAP test set over the 8 GPUs of a node):

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m astts.cli.search_json --input_json ... (same flags)

one process per GPU, the bank (and the embedder) replicated; rank r labels / embeds / searches rows [r ceil(Q/W), (r+1) ceil(Q/W)) of the
input, ONE all-gather of the retrieved (style id, similarity) pairs over RCCL (astts.parallel.sharded_search), and rank 0 writes the
JSONL in input order -- the file is identical to the one-process run's.
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


PLACEHOLDER_BIOGRAPHY = "This is a placeholder biography."      # milvus/search_json.py:375,400


def load_biographies(path):
    """{speaker: biography} from a JSON file (a dict, or a list / JSONL of {"speaker", "biography"} records as the reference's
    bank-construction dumps hold them: output_emb/embeddings_biographies_en*.json)."""
    if not path:
        return {}
    with open(path, "r", encoding="utf-8") as f:
        text = f.read().strip()
    try:
        data = json.loads(text)
    except json.JSONDecodeError:
        data = [json.loads(line) for line in text.splitlines() if line.strip()]
    if isinstance(data, dict):
        return {str(k): str(v) for k, v in data.items()}
    return {str(d["speaker"]): str(d["biography"]) for d in data if "speaker" in d and "biography" in d}


def embed_rows(rows, embedder, biographies, max_new_tokens=10, batch=32):
    """milvus/search_json.py:382-411 for a list of rows -> (queries float32 [n, 2 * hidden], labels, failed bool [n]).  The labels come
    from batched KV-cached greedy decodes (``batch`` rows at a time, sorted by prompt length so that the left padding stays small); each
    DISTINCT label / biography text is embedded once (the reference re-embeds them per row: same vectors).  Failures keep the reference's
    per-row semantics (milvus/search_json.py:389-404): a label that cannot be generated becomes "neutral", a row whose embedding fails is
    marked ``failed`` (its record says "Error") -- the other rows of the shard go on."""
    texts = [r.get("zh_text", "").strip() for r in rows]
    labels = [None] * len(rows)
    order = sorted(range(len(rows)), key=lambda i: len(texts[i]))
    for c0 in range(0, len(order), batch):
        chunk = order[c0:c0 + batch]
        try:
            labs = embedder.generate_emotion_labels([texts[i] for i in chunk], max_new_tokens)
        except Exception as e:  # noqa: BLE001
            print(f"Error during emotion generation: {e}")
            labs = ["neutral"] * len(chunk)
        for i, lab in zip(chunk, labs):
            labels[i] = lab
    bio_of = [biographies.get(r.get("speaker", "UNKNOWN_SPEAKER"), PLACEHOLDER_BIOGRAPHY) for r in rows]
    uniq = sorted(set(labels) | set(bio_of), key=len)
    vec = {}
    for c0 in range(0, len(uniq), batch):
        chunk = uniq[c0:c0 + batch]
        try:
            for t, e in zip(chunk, embedder.get_embeddings(chunk)):
                vec[t] = e
        except Exception as e:  # noqa: BLE001
            print(f"Error getting embedding: {e}")
    h = embedder.cfg.hidden
    q = np.zeros((len(rows), 2 * h), np.float32)
    failed = np.zeros(len(rows), bool)
    for i in range(len(rows)):
        if labels[i] in vec and bio_of[i] in vec:
            q[i, :h] = vec[labels[i]]                  # :263-264 concatenate((emotion_emb, bio_emb))
            q[i, h:] = vec[bio_of[i]]
        else:
            failed[i] = True
    return q, labels, failed


def main(args, client=None, embedder=None):
    import torch

    dist, rank, world, local = parallel.init_from_env()
    if dist is not None and dist.get_backend() == "nccl":
        torch.cuda.set_device(local)
    client = client or MilvusClient(args.db_path)
    rows = read_input_json(args.input_json)
    q = None
    if getattr(args, "query_npy", ""):
        q = np.load(args.query_npy).astype(np.float32)
        if q.shape[0] != len(rows):
            raise SystemExit(f"{args.query_npy}: {q.shape[0]} vectors for {len(rows)} input rows")
    # rows without text are skipped, as the reference does (milvus/search_json.py:385-387) -- together with their query vector
    keep = [i for i, r in enumerate(rows) if r.get("zh_text", "").strip()]
    if rank == 0:
        for i in sorted(set(range(len(rows))) - set(keep)):
            print(f"Skipping empty text for speaker '{rows[i].get('speaker', 'UNKNOWN_SPEAKER')}'.")
    rows = [rows[i] for i in keep]
    labels = None
    failed_local = None
    if q is not None:
        q = q[keep]
    elif len(rows):
        # the LLM half (:372-411) for THIS rank's rows only: the full-size query array is filled in the rank's shard and nowhere
        # else (sharded_search reads exactly that slice; no query vector crosses GPUs)
        b0, b1, _ = parallel.shard_bounds(len(rows), world, rank)
        with parallel.rank_work(dist, "search_json: load the embedder, emotion label + embedding"):     # agreed before the search's all-gather:
            # a rank that cannot even load its model / biographies fails its peers here instead of leaving them inside the collective
            if embedder is None:
                from astts.cli.search_milvus import load_embedder
                embedder = load_embedder(args.model_path, getattr(args, "allow_random_init", False), args.seed)
            bios = load_biographies(getattr(args, "biography_json", ""))
            q, labels, failed_local = embed_rows(rows[b0:b1], embedder, bios, max_new_tokens=10, batch=getattr(args, "llm_batch", 32))
        full = np.zeros((len(rows), q.shape[1] if len(q) else 2 * embedder.cfg.hidden), np.float32)
        full[b0:b1] = q
        q = full
        if rank == 0 and getattr(args, "verbose", False):
            for r, lab in zip(rows[b0:b1], labels):
                print(f"Emotion label for text: '{lab}'.")
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
            if failed_local is not None and failed_local.any() and len(failed_local) == idx.shape[0]:
                idx = idx.copy()
                idx[failed_local] = -2  # this rank's rows whose embedding failed: every rank learns it through the all-gather
            return torch.from_numpy(idx).to(dev), torch.from_numpy(score).to(dev)

        try:
            idx, score = parallel.sharded_search(search_fn, torch.from_numpy(q).to(dev), 1, dist)
            idx_np = idx.cpu().numpy()
            hits = client.hits_from_rows(args.collection_name, idx_np, score.cpu().numpy(), ["file_id", "text"])
            for i in np.nonzero(idx_np[:, 0] == -2)[0]:
                hits[i] = "Error"
        except Exception as e:  # noqa: BLE001
            if dist is not None:        # a rank that fails alone would leave the others inside the collective: fail the job
                raise
            print(f"Error during search: {e}")
            traceback.print_exc()
            hits = None
    for i, sample in enumerate(rows):
        zh_text = sample.get("zh_text", "").strip()
        speaker = sample.get("speaker", "UNKNOWN_SPEAKER")
        if hits is None or hits[i] == "Error":
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
    p.add_argument("--query_npy", default="", help="[rows, dim] float32 query vectors (emotion | biography halves) INSTEAD of the LLM half")
    p.add_argument("--model_path", "--llm_dir", dest="model_path", default="", help="Llama-3.2-3B checkpoint directory (merged weights + tokenizer): "
                   "the reference's --model_path (milvus/search_json.py:469)")
    p.add_argument("--biography_json", default="", help="{speaker: biography} (the reference samples these from the LLM; absent speakers "
                   "get its fallback text)")
    p.add_argument("--allow_random_init", action="store_true", help="run on seeded random Llama weights when model_path does not exist")
    p.add_argument("--llm_batch", type=int, default=32, help="rows per batched greedy decode / embedding pass")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--db_path", default="milvus_demo.db")
    p.add_argument("--collection_name", default="embeddings_biographies_collection")
    p.add_argument("--output_file", default="")
    p.add_argument("--file_prefix_path", default="")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
    parallel.shutdown()
