"""The query embedder on the GPU: Llama-3.2 decoder forward -> mean-pooled last hidden state, and the greedy
continuation that names the emotion.

Replaces, in the reference's retrieval scripts (paths under /root/reference):
    get_embedding(text, model, tokenizer, device, layer=-1, pooling='mean')   src/search_milvus.py:75-108
                                                                              (same function: milvus/search_json.py:76-109)
    generate_emotion_label(text, ...)  -> model.generate(do_sample=False)     milvus/search_json.py:154-198
    create_combined_embedding(...)     -> concatenate(emotion, biography)     milvus/search_json.py:201-229,
                                                                              src/search_milvus.py:214-221
Every tensor operation is a HIP kernel of libastts.so (GEMMs: the MFMA family of csrc/ops_gemm.hip with fp16 activations;
RMSNorm / RoPE / causal GQA attention at head_dim 128 / SwiGLU / mean-pool: csrc/ops_llm.hip).  fp16 weights and MFMA
operands, fp32 residual stream, norms and softmax -- the reference itself runs the model in fp16 with 8-bit weights
(src/search_milvus.py:47-62).  Parity: tests/test_llm_gpu.py against fixtures produced by transformers (fp32).

The tokenizer is the checkpoint's own (tokenizer.json: not available offline); anything with ``encode(text) -> list[int]``
plugs in (`transformers.AutoTokenizer` when the checkpoint directory is given).  ``HashTokenizer`` is a labelled
deterministic stand-in so that the CLIs run end to end without one.
"""
from __future__ import annotations

import threading

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from ..ops import PackedWeight
from .config import LlamaShape


def llama3_inv_freq(cfg: LlamaShape) -> torch.Tensor:
    """transformers' _compute_llama3_parameters, float32 as there."""
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.int64).float() / cfg.head_dim))
    low_wl = cfg.rope_original_max_pos / cfg.rope_low_freq_factor
    high_wl = cfg.rope_original_max_pos / cfg.rope_high_freq_factor
    wl = 2 * math.pi / inv
    inv_l = torch.where(wl > low_wl, inv / cfg.rope_factor, inv)
    smooth = (cfg.rope_original_max_pos / wl - cfg.rope_low_freq_factor) / (cfg.rope_high_freq_factor - cfg.rope_low_freq_factor)
    smoothed = (1 - smooth) * inv_l / cfg.rope_factor + smooth * inv_l
    medium = ~(wl < high_wl) * ~(wl > low_wl)
    return torch.where(medium, smoothed, inv_l)


class HashTokenizer:
    """STAND-IN (the Llama tokenizer files do not exist offline): bos + one id per whitespace-separated word by a fixed
    hash.  Deterministic, reversible in nothing; good for plumbing and benchmarks only."""

    def __init__(self, cfg: LlamaShape):
        self.cfg = cfg

    def encode(self, text: str) -> List[int]:
        import zlib

        return [self.cfg.bos_token_id] + [3 + zlib.crc32(w.encode("utf-8")) % (self.cfg.vocab - 3) for w in text.split()]

    def decode(self, ids: Sequence[int]) -> str:
        return " ".join(f"<{int(i)}>" for i in ids)


class LlamaEmbedder:
    def __init__(self, state: dict, cfg: LlamaShape, device=None, tokenizer=None, max_length: int = 512):
        if not torch.cuda.is_available():
            raise RuntimeError("astts.llm needs a ROCm GPU; there is no CPU fallback in the product path")
        if cfg.head_dim != 128:
            raise ValueError("LlamaEmbedder: the attention kernel is built for head_dim 128 (Llama-3.2)")
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.tokenizer = tokenizer or HashTokenizer(cfg)
        self.max_length = max_length                                  # truncation=True, max_length=512: src/search_milvus.py:92
        dev = self.device
        with torch.cuda.device(dev):
            f = lambda k: state[k].to(device=dev, dtype=torch.float32).contiguous()
            self.embed = f("model.embed_tokens.weight")               # fp32 table: the lookup feeds the fp32 residual stream
            self.L = []
            for i in range(cfg.layers):
                p = f"model.layers.{i}."
                wqkv = torch.cat([state[p + "self_attn.q_proj.weight"], state[p + "self_attn.k_proj.weight"],
                                  state[p + "self_attn.v_proj.weight"]], 0)
                wgu = torch.cat([state[p + "mlp.gate_proj.weight"], state[p + "mlp.up_proj.weight"]], 0)
                self.L.append({"n1": f(p + "input_layernorm.weight"), "n2": f(p + "post_attention_layernorm.weight"),
                               "wqkv": PackedWeight(wqkv, None, dev), "wo": PackedWeight(state[p + "self_attn.o_proj.weight"], None, dev),
                               "wgu": PackedWeight(wgu, None, dev), "wd": PackedWeight(state[p + "mlp.down_proj.weight"], None, dev)})
            self.norm = f("model.norm.weight")
            head = state["model.embed_tokens.weight"] if cfg.tie_embeddings else state["lm_head.weight"]
            self.head = PackedWeight(head, None, dev)
            self._rope_lock = threading.Lock()
            self._rope_tables(max(max_length, 16) + 64)

    def _rope_tables(self, n: int) -> None:
        """(cos, sin) rows for positions < n, published as ONE tuple: a thread that sees the new cos also sees the new sin."""
        fr = torch.arange(n, dtype=torch.float32)[:, None] * llama3_inv_freq(self.cfg)[None, :]
        self._rope = (fr.cos().to(self.device).contiguous(), fr.sin().to(self.device).contiguous())

    @property
    def cos(self) -> torch.Tensor:
        return self._rope[0]

    @property
    def sin(self) -> torch.Tensor:
        return self._rope[1]

    # ------------------------------------------------------------------ the decoder stack
    def hidden(self, ids: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids int [B, T] (right-padded), lens int32 [B] or None -> final-norm hidden states fp32 [B, T, hidden]
        (== outputs.hidden_states[-1] of LlamaModel)."""
        cfg = self.cfg
        b, t = ids.shape
        if t > self._rope[0].shape[0]:  # the untruncated generation prompt (milvus/search_json.py:178) can exceed max_length
            with self._rope_lock:
                if t > self._rope[0].shape[0]:
                    self._rope_tables((t + 255) // 256 * 256)
        cos, sin = self._rope           # one consistent pair for the whole pass
        hq, hk = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
        x = ops.embedding(self.embed, ids.to(self.device))
        for L in self.L:
            h = ops.rmsnorm(x, L["n1"], cfg.rms_eps)                              # fp16: its only consumer is an MFMA operand
            qkv = ops.linear(h, L["wqkv"], out_dtype=torch.float16)              # [B, T, hq + 2 hk]
            ops.rope_llama_(qkv, cos, sin, cfg.heads + cfg.kv_heads, cfg.head_dim)             # q heads then k heads: contiguous
            a = ops.attn_causal_gqa(qkv[..., :hq], qkv[..., hq:hq + hk], qkv[..., hq + hk:], cfg.heads, cfg.kv_heads, cfg.head_dim, lens)
            x = ops.linear(a, L["wo"], residual=x)
            h = ops.rmsnorm(x, L["n2"], cfg.rms_eps)
            gu = ops.linear(h, L["wgu"], out_dtype=torch.float16)
            x = ops.linear(ops.swiglu(gu), L["wd"], residual=x)
        return ops.rmsnorm(x, self.norm, cfg.rms_eps, out_dtype=torch.float32)

    def embed_ids(self, ids: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Mean over each text's tokens of the last hidden state -> fp32 [B, hidden] on the GPU."""
        if lens is not None:
            lens = lens.to(device=self.device, dtype=torch.int32)
        return ops.mean_pool(self.hidden(ids, lens), lens)

    def logits_last(self, ids: torch.Tensor) -> torch.Tensor:
        h = self.hidden(ids)[:, -1]
        return ops.linear(h.contiguous(), self.head)

    def generate_greedy(self, ids: Sequence[int], max_new_tokens: int = 10) -> List[int]:
        """do_sample=False continuation (milvus/search_json.py:178-188).  The prompt is re-run for every new token: the
        label is <= 10 tokens behind a ~60-token prompt, and the embedding passes -- not this -- are the volume."""
        out = list(int(i) for i in ids)
        for _ in range(max_new_tokens):
            lg = self.logits_last(torch.tensor([out], dtype=torch.int64, device=self.device))
            nxt = int(torch.argmax(lg[0]))                                        # one scalar back to the host per token
            out.append(nxt)
            if nxt == self.cfg.eos_token_id:
                break
        return out

    # ------------------------------------------------------------------ the reference's call surface
    def _encode(self, text: str) -> List[int]:
        ids = list(self.tokenizer.encode(text))
        return ids[: self.max_length]

    def get_embedding(self, text: str) -> np.ndarray:
        """src/search_milvus.py:75-108 with layer=-1, pooling='mean' -> numpy float32 [hidden]."""
        ids = torch.tensor([self._encode(text)], dtype=torch.int64)
        return self.embed_ids(ids).cpu().numpy()[0]

    def get_embeddings(self, texts: Sequence[str]) -> np.ndarray:
        """Many texts in one right-padded batch; each row equals get_embedding(text) (padding is masked)."""
        enc = [self._encode(t) for t in texts]
        tmax = max(len(e) for e in enc)
        ids = torch.zeros((len(enc), tmax), dtype=torch.int64)
        for i, e in enumerate(enc):
            ids[i, : len(e)] = torch.tensor(e)
        return self.embed_ids(ids, torch.tensor([len(e) for e in enc])).cpu().numpy()

    EMOTION_PROMPT = """\n=======
Context: Given predefined emotional label set [happy, sad, neutral, angry, excited, frustrated], and below conversation:
"
{}
"

Question: What is the emotion of the speaker at the utterance "{}"?
Answer:"""

    def generate_emotion_label(self, text: str, max_new_tokens: int = 10) -> str:
        """milvus/search_json.py:154-198: greedy continuation of the few-shot prompt, decoded (prompt included, as there),
        stripped and lower-cased."""
        prompt = self.EMOTION_PROMPT.format(text, text)
        out = self.generate_greedy(list(self.tokenizer.encode(prompt)), max_new_tokens)     # untruncated: only get_embedding truncates there
        label = self.tokenizer.decode(out, skip_special_tokens=True) if self._decode_takes_skip else self.tokenizer.decode(out)   # search_json.py:191
        return label.strip().lower()

    @property
    def _decode_takes_skip(self) -> bool:
        """Decided from the tokenizer's signature, once: a TypeError raised INSIDE a real tokenizer's decode must not be mistaken for
        "this stand-in has no skip_special_tokens argument" (and silently decoded with the special tokens in)."""
        if not hasattr(self, "_decode_skip"):
            import inspect
            try:
                ps = inspect.signature(self.tokenizer.decode).parameters
                self._decode_skip = "skip_special_tokens" in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values())
            except (TypeError, ValueError):
                self._decode_skip = True
        return self._decode_skip

    def combined_embedding(self, emotion_text: str, biography_text: str) -> np.ndarray:
        """milvus/search_json.py:201-229 / src/search_milvus.py:214-221: [emotion | biography] float32, un-normalised."""
        e = self.get_embeddings([emotion_text, biography_text])
        return np.concatenate((e[0], e[1])).astype(np.float32)
