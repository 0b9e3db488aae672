"""Restatement of /root/reference/milvus/search_embeddings.py: same flags, same printed lines, same error
convention (the wrapper swallows search errors: print + traceback, return [] -- :24-27), served by the HBM-resident
style bank through the MilvusClient shim.

    python -m astts.cli.search_embeddings --query_embedding q.json --top_k 3 --db_path milvus_demo.db
"""
import argparse
import json
import traceback

from astts.compat.pymilvus import MilvusClient


def search_milvus(client, collection_name, embedding, top_k=3):
    try:
        param = {"nprobe": 10}
        return client.search(collection_name=collection_name, data=[embedding], anns_field="vector", param=param,
                             limit=top_k, output_fields=["file_id", "text"])
    except Exception as e:  # noqa: BLE001 -- the reference degrades to [] on any failure
        print(f"Error during Milvus search: {e}")
        traceback.print_exc()
        return []


def main(args):
    client = MilvusClient(args.db_path)
    print(f"Connected to Milvus at '{args.db_path}'.")
    try:
        with open(args.query_embedding, "r", encoding="utf-8") as f:
            query_embedding = json.load(f)
        print(f"Loaded query embedding from '{args.query_embedding}'.")
    except Exception as e:  # noqa: BLE001
        print(f"Error loading query embedding: {e}")
        traceback.print_exc()
        return []
    search_results = search_milvus(client, "embeddings_biographies_collection", query_embedding, top_k=args.top_k)
    if search_results:
        for i, result in enumerate(search_results):
            print(f"\nTop {args.top_k} results for Query {i + 1}:")
            for res in result:
                print(f"File ID: {res['entity']['file_id']}, Distance: {res['distance']}, Text: {res['entity']['text']}")
            print("-" * 50)
    else:
        print("No results found.")
    return search_results


def build_parser():
    parser = argparse.ArgumentParser(description="Search embeddings in Milvus Lite")
    parser.add_argument("--query_embedding", type=str, required=True,
                        help="Path to the JSON file containing the query embedding vector")
    parser.add_argument("--top_k", type=int, default=3, help="Number of top similar results to retrieve")
    parser.add_argument("--db_path", type=str, default="milvus_demo.db", help="Path to the Milvus Lite database file")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
