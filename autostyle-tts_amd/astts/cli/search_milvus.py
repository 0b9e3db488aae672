"""Restatement of /root/reference/src/search_milvus.py (:126-262): embed a query text and a speaker biography with the
Llama embedder, concatenate them into the 6144-d style-bank query, search the COSINE collection, print the hits.  Same
flags, same printed lines, same error convention (every stage prints + returns on failure; the search wrapper swallows to
[] -- :149-152).  The model is this build's GPU embedder (astts.llm.embedder.LlamaEmbedder) instead of an 8-bit PEFT
Llama under transformers (:36-72); a ready embedder can be injected (``main(args, embedder=...)``).

    python -m astts.cli.search_milvus --db_path milvus_demo.db --model_path /path/to/llama-3.2-3b \\
        --search_text "A man with a humorous style" --query_speaker ELIZABETH --top_k 3
"""
import argparse
import traceback

import numpy as np

from astts.compat.pymilvus import MilvusClient

# src/search_milvus.py:111-116
speaker_bio = {
    "ELIZABETH": "Elizabeth is a deeply emotional and passionate individual, extremely devoted to her husband Larry, willing to risk her life to hold onto their relationship.",
    "JOHN": "John is a pragmatic and thoughtful person, always considering the practical aspects of any situation. He values honesty and reliability in his relationships.",
    "PATRICIA": "Patricia is a thoughtful and reserved individual who values her privacy and independence. Her interactions reveal a sense of determination and a desire for meaningful connections.",
    "ANN": "Ann is a cheerful and optimistic person, always looking for the silver lining in every situation. She enjoys connecting with others and values strong interpersonal relationships.",
}


def load_embedder(model_path, allow_random_init=False, seed=42):
    """:36-72 ``load_model_and_tokenizer``: the checkpoint directory must exist (safetensors / .bin shards + tokenizer).
    Without one this raises unless random-init weights are explicitly allowed (plumbing runs, benchmarks)."""
    import os

    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import load_llama_weights, make_llama_weights

    cfg = LlamaShape.llama32_3b()
    if model_path and os.path.isdir(model_path):
        state = load_llama_weights(model_path)
        tok = None
        try:  # the checkpoint's own tokenizer when transformers can read it
            from transformers import AutoTokenizer

            tok = AutoTokenizer.from_pretrained(model_path)
        except Exception as e:  # noqa: BLE001
            print(f"Warning: no tokenizer under '{model_path}' ({e}); using the hash stand-in")
        return LlamaEmbedder(state, cfg, tokenizer=tok)
    if not (allow_random_init or os.environ.get("ASTTS_ALLOW_RANDOM_INIT") == "1"):
        raise FileNotFoundError(f"no checkpoint directory at {model_path!r} (pass --allow_random_init to run on seeded random weights)")
    if os.environ.get("ASTTS_TINY_MODEL") == "1":
        cfg = LlamaShape.tiny()
    print(f"Warning: '{model_path}' not found; seeded RANDOM-INIT Llama weights at {cfg.hidden}-d (explicitly allowed)")
    return LlamaEmbedder(make_llama_weights(cfg, seed), cfg)


def emb_text_bio(speaker, embedder):
    """:118-123"""
    return embedder.get_embedding(speaker_bio.get(speaker.upper(), "unknown"))


def search_milvus(client, collection_name, embedding, top_k=3):
    """:126-152"""
    try:
        return client.search(collection_name=collection_name, data=[embedding], anns_field="vector", metric_type="COSINE",
                             limit=top_k, output_fields=["file_id"])
    except Exception as e:  # noqa: BLE001
        print(f"Error during Milvus search: {e}")
        traceback.print_exc()
        return []


def main(args, embedder=None):
    if embedder is None:
        embedder = load_embedder(args.model_path, getattr(args, "allow_random_init", False), args.seed)
    try:
        client = MilvusClient(args.db_path)
        print(f"Connected to Milvus database at '{args.db_path}'.")
    except Exception as e:  # noqa: BLE001
        print(f"Error connecting to Milvus: {e}")
        traceback.print_exc()
        return None
    collection_name = args.collection_name
    if not client.has_collection(collection_name=collection_name):
        print(f"Collection '{collection_name}' does not exist. Please check the collection name.")
        return None
    try:
        collection_info = client.get_collection_info(collection_name)
        print(f"Collection '{collection_name}' info:")
        print(collection_info)
    except Exception as e:  # noqa: BLE001
        print(f"Error retrieving collection info: {e}")
        traceback.print_exc()
        return None
    query_text = args.search_text
    try:
        emotion_emb = embedder.get_embedding(query_text)                     # :214
        bio_emb = emb_text_bio(args.query_speaker, embedder)                 # :217
        combined_emb = np.concatenate((emotion_emb, bio_emb))                # :220
        combined_emb = combined_emb.astype(np.float32).tolist()
        print(f"Generated combined embedding of shape {len(combined_emb)}.")
    except Exception as e:  # noqa: BLE001
        print(f"Error generating embedding for the query text: {e}")
        traceback.print_exc()
        return None
    search_results = search_milvus(client, collection_name, combined_emb, top_k=args.top_k)
    if search_results:
        for query_idx, query_result in enumerate(search_results):
            print(f"\nTop {args.top_k} results for the query '{query_text}':")
            for res in query_result:
                print(f"File ID: {res['entity']['file_id']}, Distance: {res['distance']}")
            print("-" * 50)
    else:
        print("No results found.")
    return search_results


def build_parser():
    parser = argparse.ArgumentParser(description="Search embeddings in Milvus Lite")
    parser.add_argument("--db_path", type=str, default="milvus_demo.db", help="Path to the Milvus Lite database file")
    parser.add_argument("--collection_name", type=str, default="embeddings_biographies_collection",
                        help="Name of the Milvus collection to search")
    parser.add_argument("--model_path", type=str, default="", help="Path to the fine-tuned Llama 3.2 model (merged weights)")
    parser.add_argument("--allow_random_init", action="store_true",
                        help="run on seeded random weights when model_path does not exist (otherwise that is an error)")
    parser.add_argument("--search_text", type=str,
                        default="A man with a humorous style, from the countryside, with a very delicate mind",
                        help="Text to perform search in Milvus")
    parser.add_argument("--query_speaker", type=str, default="ELIZABETH", help="Speaker associated with the search text")
    parser.add_argument("--top_k", type=int, default=3, help="Number of top similar results to retrieve")
    parser.add_argument("--seed", type=int, default=42, help="Random seed for reproducibility")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
