"""CPU oracle for the retrieval path's query embedder.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What the reference computes (/root/reference/src/search_milvus.py:75-108, milvus/search_json.py:154-198,214-221):
    get_embedding(text)      = mean over tokens of outputs.hidden_states[-1] of model.model(input_ids, attention_mask,
                               output_hidden_states=True), cast fp16 -> fp32 (3072-d for Llama-3.2-3B);
    generate_emotion_label() = model.generate(prompt, max_new_tokens=10, do_sample=False): greedy continuation;
    combined                 = concatenate(emotion_emb, bio_emb)  -> the 6144-d style-bank query.
The arithmetic lives in the third-party ``transformers`` package (LlamaModel; no version pinned in the reference tree).
This file restates LlamaModel / LlamaForCausalLM.forward in fp32 torch, following transformers'
models/llama/modeling_llama.py and modeling_rope_utils.py::_compute_llama3_parameters.

PARITY PINNED: tests/golden/make_llama_fixtures.py imports transformers 5.15 (present in the build image), loads the
seeded weights of astts.llm.weights.make_llama_weights into LlamaForCausalLM and records hidden states, pooled embeddings
and greedy tokens; tests/test_oracle_llama.py holds this restatement to those fixtures (1e-5).
The reference itself runs the model 8-bit-quantised in fp16 (src/search_milvus.py:47-62): not reproducible bit for bit by
anything but bitsandbytes; the fp32 definition is the ground truth both are approximations of.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

SD = Dict[str, torch.Tensor]


def llama3_inv_freq(head_dim: int, theta: float, factor: float, low_freq_factor: float, high_freq_factor: float,
                    original_max_pos: int) -> torch.Tensor:
    """modeling_rope_utils._compute_llama3_parameters (float32 throughout, as transformers)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    low_freq_wavelen = original_max_pos / low_freq_factor
    high_freq_wavelen = original_max_pos / high_freq_factor
    wavelen = 2 * math.pi / inv_freq
    inv_freq_llama = torch.where(wavelen > low_freq_wavelen, inv_freq / factor, inv_freq)
    smooth = (original_max_pos / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor)
    smoothed = (1 - smooth) * inv_freq_llama / factor + smooth * inv_freq_llama
    is_medium = ~(wavelen < high_freq_wavelen) * ~(wavelen > low_freq_wavelen)
    return torch.where(is_medium, smoothed, inv_freq_llama)


def rope_tables(cfg, t: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos / sin [t, head_dim] as LlamaRotaryEmbedding.forward builds them: emb = cat(freqs, freqs)."""
    inv = llama3_inv_freq(cfg.head_dim, cfg.rope_theta, cfg.rope_factor, cfg.rope_low_freq_factor, cfg.rope_high_freq_factor,
                          cfg.rope_original_max_pos)
    freqs = torch.arange(t, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def forward_hidden(sd: SD, cfg, ids: torch.Tensor, lens: torch.Tensor = None, all_layers: bool = False):
    """ids int64 [B, T] (right-padded; ``lens`` [B] valid tokens, None = all) -> last hidden state after the final norm
    [B, T, hidden] (== outputs.hidden_states[-1]); with all_layers also the list of per-layer inputs."""
    b, t = ids.shape
    x = sd["model.embed_tokens.weight"][ids]
    cos, sin = rope_tables(cfg, t)
    neg = torch.full((t, t), float("-inf")).triu(1)                              # causal
    mask = neg[None, None].expand(b, 1, t, t).clone()
    if lens is not None:
        pad = torch.arange(t)[None, :] >= lens[:, None]                          # padded keys
        mask = mask.masked_fill(pad[:, None, None, :], float("-inf"))
    hs: List[torch.Tensor] = []
    rep = cfg.heads // cfg.kv_heads
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        hs.append(x)
        h = rmsnorm(x, sd[p + "input_layernorm.weight"], cfg.rms_eps)
        q = (h @ sd[p + "self_attn.q_proj.weight"].T).view(b, t, cfg.heads, cfg.head_dim).transpose(1, 2)
        k = (h @ sd[p + "self_attn.k_proj.weight"].T).view(b, t, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        v = (h @ sd[p + "self_attn.v_proj.weight"].T).view(b, t, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
        s = q @ k.transpose(-1, -2) / math.sqrt(cfg.head_dim) + mask
        a = torch.softmax(s, dim=-1, dtype=torch.float32) @ v
        a = a.transpose(1, 2).reshape(b, t, cfg.heads * cfg.head_dim)
        x = x + a @ sd[p + "self_attn.o_proj.weight"].T
        h = rmsnorm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_eps)
        g = h @ sd[p + "mlp.gate_proj.weight"].T
        u = h @ sd[p + "mlp.up_proj.weight"].T
        x = x + (torch.nn.functional.silu(g) * u) @ sd[p + "mlp.down_proj.weight"].T
    out = rmsnorm(x, sd["model.norm.weight"], cfg.rms_eps)
    return (out, hs) if all_layers else out


def get_embedding(sd: SD, cfg, ids: torch.Tensor, lens: torch.Tensor = None) -> torch.Tensor:
    """src/search_milvus.py:75-108 with pooling='mean', layer=-1: mean over the (valid) tokens of the final hidden state.
    (The reference embeds one text at a time, so its plain .mean(dim=1) never sees padding.)"""
    h = forward_hidden(sd, cfg, ids, lens)
    if lens is None:
        return h.mean(dim=1)
    m = (torch.arange(ids.shape[1])[None, :] < lens[:, None]).float()[..., None]
    return (h * m).sum(1) / lens[:, None].float()


def logits_last(sd: SD, cfg, ids: torch.Tensor) -> torch.Tensor:
    h = forward_hidden(sd, cfg, ids)[:, -1]
    w = sd["model.embed_tokens.weight"] if cfg.tie_embeddings else sd["lm_head.weight"]
    return h @ w.T


def generate_greedy(sd: SD, cfg, ids: torch.Tensor, max_new_tokens: int) -> torch.Tensor:
    """milvus/search_json.py:178-188: do_sample=False, stop at eos.  ids [1, T] -> [1, T + n]."""
    out = ids.clone()
    for _ in range(max_new_tokens):
        nxt = int(torch.argmax(logits_last(sd, cfg, out)[0]))
        out = torch.cat([out, torch.tensor([[nxt]])], dim=1)
        if nxt == cfg.eos_token_id:
            break
    return out


def combined_query(emotion_emb: torch.Tensor, bio_emb: torch.Tensor) -> torch.Tensor:
    """src/search_milvus.py:220-221 / milvus/search_json.py:226: concat(emotion, biography) as float32."""
    return torch.cat([emotion_emb.reshape(-1), bio_emb.reshape(-1)]).float()
