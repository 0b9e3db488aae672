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
        import os
        self.mfma_attention = os.environ.get("ASTTS_LLM_ATTN", "mfma") != "valu"
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
            if self.mfma_attention:                                                # v_mfma_f32_32x32x16_f16 (csrc/ops_llm.hip attn_gqa_mfma)
                a = ops.attn_gqa(qkv[..., :hq], qkv[..., hq:hq + hk], qkv[..., hq + hk:], cfg.heads, cfg.kv_heads, cfg.head_dim, lens=lens)
            else:                                                                  # the VALU kernel: the second implementation (tests)
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

    def generate_greedy_recompute(self, ids: Sequence[int], max_new_tokens: int = 10) -> List[int]:
        """do_sample=False continuation with the prompt re-run for every new token (rounds 3-4; kept as the second implementation
        the tests hold the cached path to)."""
        out = list(int(i) for i in ids)
        for _ in range(max_new_tokens):
            lg = self.logits_last(torch.tensor([out], dtype=torch.int64, device=self.device))
            nxt = int(torch.argmax(lg[0]))                                        # one scalar back to the host per token
            out.append(nxt)
            if nxt == self.cfg.eos_token_id:
                break
        return out

    def generate_greedy_batch(self, prompts: Sequence[Sequence[int]], max_new_tokens: int = 10) -> List[List[int]]:
        """do_sample=False continuation (milvus/search_json.py:178-188: model.generate(max_new_tokens=10)) of several prompts at once,
        with a KV cache and no host synchronisation inside the loop: ONE pass over the prompts, then ``max_new_tokens - 1`` one-token
        steps; the argmax runs on the device (``astts_op_argmax_rows``) and feeds the next step's embedding lookup; the tokens come
        back in one copy at the end and are cut at each row's first EOS on the host (transformers stops a row there).
        Layout: prompts are LEFT-padded to a common length, time-major ``[T, B]`` (the rows of a step are contiguous in the cache
        ``[T_max, B, 2 * kv_heads * 128]`` per layer); ``key_start[b]`` masks a row's pad keys and shifts its RoPE positions so that its
        first token has position 0, as in the one-at-a-time reference run."""
        cfg, dev = self.cfg, self.device
        prompts = [[int(i) for i in p] for p in prompts]
        b, lens = len(prompts), [len(p) for p in prompts]
        t, n_new = max(lens), int(max_new_tokens)
        if n_new <= 0:
            return [list(p) for p in prompts]
        t_max = t + n_new
        if t_max > self._rope[0].shape[0]:
            with self._rope_lock:
                if t_max > self._rope[0].shape[0]:
                    self._rope_tables((t_max + 255) // 256 * 256)
        cos, sin = self._rope
        hq, hk = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
        ids = torch.zeros((t, b), dtype=torch.int32)
        for j, p in enumerate(prompts):
            ids[t - lens[j]:, j] = torch.tensor(p, dtype=torch.int32)
        start = torch.tensor([t - n for n in lens], dtype=torch.int32, device=dev)
        cache = [torch.empty((t_max, b, 2 * hk), dtype=torch.float16, device=dev) for _ in self.L]
        toks = torch.zeros((n_new, b), dtype=torch.int32, device=dev)

        def stack(x: torch.Tensor, pos0: int) -> torch.Tensor:
            """x fp32 [T', B, hidden] = the new positions pos0 .. pos0 + T' - 1 -> final-norm hidden of the LAST of them [B, hidden]."""
            tn = x.shape[0]
            for L, kv in zip(self.L, cache):
                h = ops.rmsnorm(x, L["n1"], cfg.rms_eps)
                qkv = ops.linear(h, L["wqkv"], out_dtype=torch.float16)                       # [T', B, hq + 2 hk]
                ops.rope_llama_ex_(qkv, cos, sin, cfg.heads + cfg.kv_heads, cfg.head_dim, pos0=pos0, shift=start, time_major=True)
                kv[pos0:pos0 + tn].copy_(qkv[..., hq:])                                        # K (rotated) | V into the cache rows
                a = ops.attn_gqa(qkv[..., :hq], kv[:pos0 + tn, :, :hk], kv[:pos0 + tn, :, hk:], cfg.heads, cfg.kv_heads, cfg.head_dim,
                                 key_start=start, pos0=pos0, time_major=True)
                x = ops.linear(a, L["wo"], residual=x)
                h = ops.rmsnorm(x, L["n2"], cfg.rms_eps)
                gu = ops.linear(h, L["wgu"], out_dtype=torch.float16)
                x = ops.linear(ops.swiglu(gu), L["wd"], residual=x)
            return ops.rmsnorm(x[-1].contiguous(), self.norm, cfg.rms_eps, out_dtype=torch.float32)

        x = ops.embedding(self.embed, ids.to(dev))
        for s in range(n_new):
            h_last = stack(x, 0 if s == 0 else t + s - 1)
            ops.argmax_rows(ops.linear(h_last, self.head), out=toks[s])
            if s + 1 < n_new:
                x = ops.embedding(self.embed, toks[s])[None]
        got = toks.cpu().numpy()                                                               # the one synchronisation
        out = []
        for j, p in enumerate(prompts):
            row = list(p)
            for s in range(n_new):
                row.append(int(got[s, j]))
                if row[-1] == cfg.eos_token_id:
                    break
            out.append(row)
        return out

    def generate_greedy(self, ids: Sequence[int], max_new_tokens: int = 10) -> List[int]:
        """do_sample=False continuation of one prompt (milvus/search_json.py:178-188): the cached path with one row."""
        return self.generate_greedy_batch([ids], max_new_tokens)[0]

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
        return self.generate_emotion_labels([text], max_new_tokens)[0]

    def generate_emotion_labels(self, texts: Sequence[str], max_new_tokens: int = 10) -> List[str]:
        """The labels of several utterances in one batched greedy decode (each equals generate_emotion_label of its text)."""
        prompts = [list(self.tokenizer.encode(self.EMOTION_PROMPT.format(t, t))) for t in texts]    # untruncated: only get_embedding truncates there
        outs = self.generate_greedy_batch(prompts, max_new_tokens)
        dec = (lambda o: self.tokenizer.decode(o, skip_special_tokens=True)) if self._decode_takes_skip else self.tokenizer.decode   # search_json.py:191
        return [dec(o).strip().lower() for o in outs]

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
