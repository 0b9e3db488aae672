"""State dicts of the embedder LLM: seeded synthetic weights at a given shape (no Llama checkpoint exists offline) and the
loader for a real one.  Names follow transformers' LlamaForCausalLM so a real checkpoint loads through the same path."""
from __future__ import annotations

import glob
import os
from typing import Dict

import torch

from .config import LlamaShape

StateDict = Dict[str, torch.Tensor]


def make_llama_weights(cfg: LlamaShape, seed: int = 0, device=None) -> StateDict:
    """Deterministic (torch.Generator, CPU) fp32 weights: projections N(0, 0.02^2 * 4) so that activations stay O(1)
    through a few layers, norm scales 1 + 0.1 N(0,1).  The fixture generator, the oracle tests and the GPU tests all
    call this -- the weights never travel, only the seed does.  ``device``: draw on that device instead (its own generator, so
    OTHER values than the CPU stream's: for throughput runs of the full 3.2 B-parameter model, where no fixture is compared)."""
    g = torch.Generator(device=device).manual_seed(seed) if device is not None else torch.Generator().manual_seed(seed)
    r = lambda *s, std: torch.randn(*s, generator=g, device=device) * std
    sd: StateDict = {"model.embed_tokens.weight": r(cfg.vocab, cfg.hidden, std=0.5)}
    kv = cfg.kv_heads * cfg.head_dim
    q = cfg.heads * cfg.head_dim
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = r(q, cfg.hidden, std=0.04)
        sd[p + "self_attn.k_proj.weight"] = r(kv, cfg.hidden, std=0.04)
        sd[p + "self_attn.v_proj.weight"] = r(kv, cfg.hidden, std=0.04)
        sd[p + "self_attn.o_proj.weight"] = r(cfg.hidden, q, std=0.02)
        sd[p + "mlp.gate_proj.weight"] = r(cfg.ffn, cfg.hidden, std=0.04)
        sd[p + "mlp.up_proj.weight"] = r(cfg.ffn, cfg.hidden, std=0.04)
        sd[p + "mlp.down_proj.weight"] = r(cfg.hidden, cfg.ffn, std=0.02)
        sd[p + "input_layernorm.weight"] = 1.0 + r(cfg.hidden, std=0.1)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + r(cfg.hidden, std=0.1)
    sd["model.norm.weight"] = 1.0 + r(cfg.hidden, std=0.1)
    if not cfg.tie_embeddings:
        sd["lm_head.weight"] = r(cfg.vocab, cfg.hidden, std=0.05)
    return sd


def load_llama_weights(model_dir: str) -> StateDict:
    """A transformers checkpoint directory (``*.safetensors`` shards or ``pytorch_model*.bin``), e.g. the fine-tuned
    Llama-3.2-3B the reference points at (src/search_milvus.py:251).  PEFT adapters must be merged beforehand."""
    sd: StateDict = {}
    shards = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if shards:
        from safetensors.torch import load_file

        for s in shards:
            sd.update(load_file(s, device="cpu"))
    else:
        bins = sorted(glob.glob(os.path.join(model_dir, "pytorch_model*.bin")))
        if not bins:
            raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {model_dir!r}")
        for b in bins:
            sd.update(torch.load(b, map_location="cpu", weights_only=True))
    return {k: v.float() for k, v in sd.items()}
