"""Query embedder of the retrieval path: the Llama-3.2-3B decoder whose mean-pooled last hidden
state is one 3072-d half of a style-bank query (/root/reference/src/search_milvus.py:75-108,214-221) and whose greedy
continuation is the emotion label (/root/reference/milvus/search_json.py:154-198)."""
