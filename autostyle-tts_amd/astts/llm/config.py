"""Shapes of the embedder LLM.  The reference loads "Llama 3.2" 3B (src/search_milvus.py:251-252, milvus/RAG.py:458:
hidden 3072 -> 2 x 3072 = the 6144-d bank); these are that checkpoint's config.json values."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class LlamaShape:
    vocab: int = 128256
    hidden: int = 3072
    layers: int = 28
    heads: int = 24
    kv_heads: int = 8
    head_dim: int = 128
    ffn: int = 8192
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    # rope_scaling {"rope_type": "llama3"}
    rope_factor: float = 32.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_pos: int = 8192
    max_positions: int = 131072
    tie_embeddings: bool = True
    eos_token_id: int = 128001
    bos_token_id: int = 128000

    @staticmethod
    def llama32_3b() -> "LlamaShape":
        return LlamaShape()

    @staticmethod
    def tiny() -> "LlamaShape":
        """Small everywhere except the head dimension (128, as the real model: the attention kernel is built for it)."""
        return LlamaShape(vocab=512, hidden=512, layers=3, heads=4, kv_heads=2, ffn=1024, eos_token_id=2, bos_token_id=1)

    @staticmethod
    def wide() -> "LlamaShape":
        """The real widths (hidden 3072, 24 / 8 heads, FFN 8192) with few layers and a small vocabulary: the kernels at the
        shapes Llama-3.2-3B runs them at, in a model whose weights regenerate from a seed in seconds."""
        return LlamaShape(vocab=4096, layers=3, eos_token_id=2, bos_token_id=1)

    def hf_kwargs(self) -> dict:
        """transformers.LlamaConfig arguments (used only by tests/golden/make_llama_fixtures.py)."""
        return dict(vocab_size=self.vocab, hidden_size=self.hidden, intermediate_size=self.ffn, num_hidden_layers=self.layers,
                    num_attention_heads=self.heads, num_key_value_heads=self.kv_heads, head_dim=self.head_dim,
                    max_position_embeddings=self.max_positions, rms_norm_eps=self.rms_eps, rope_theta=self.rope_theta,
                    rope_scaling={"factor": self.rope_factor, "high_freq_factor": self.rope_high_freq_factor,
                                  "low_freq_factor": self.rope_low_freq_factor,
                                  "original_max_position_embeddings": self.rope_original_max_pos, "rope_type": "llama3"},
                    tie_word_embeddings=self.tie_embeddings, attention_bias=False, mlp_bias=False, hidden_act="silu",
                    eos_token_id=self.eos_token_id, bos_token_id=self.bos_token_id, pad_token_id=None)
