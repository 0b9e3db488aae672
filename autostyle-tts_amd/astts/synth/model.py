"""Synthesis engine: acoustic transformer -> flow-matching decoder -> HiFT vocoder on one MI355X.

This is the arithmetic the reference reaches through ``cosyvoice.inference_tts_with_st``
(/root/reference/tts_with_rag.py:195, tts_with_style_and_timbre.py:93), ``inference_zero_shot``
(:133 / :47) and ``inference_vc`` (:141 / :57).  Every tensor operation below is a hand-written
HIP kernel reached through astts.ops (C ABI); torch only owns the HBM buffers, views and
concatenations.  Activations are fp32 channels-last, weights fp16 (packed once at load), all
contractions run on fp16 MFMA with fp32 accumulation.

Randomness is injected by the caller (sampling uniforms, CFM start noise, source phases/noise) so
that the CPU oracle (oracle/synth.py) can be driven with identical draws.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

from .. import ops
from ..ops import PackedWeight
from .config import SynthConfig

SD = Dict[str, torch.Tensor]


def _dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def rel_pos_table(d: int, max_pos: int) -> torch.Tensor:
    """Constant sinusoid table (host-side constant generation): rows rel = -max_pos..max_pos."""
    rel = torch.arange(-max_pos, max_pos + 1, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * max_pos + 1, d)
    pe[:, 0::2] = torch.sin(rel * div)
    pe[:, 1::2] = torch.cos(rel * div)
    return pe


def fold_layernorm(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Linear(LayerNorm(x; gamma, beta)) == Linear'((x - mean) * rstd) with W' = W diag(gamma), b' = b + W beta
    (exact algebra, evaluated in fp64 at load): the decode step then normalises without reading gamma / beta --
    eight waves of every workgroup re-read both vectors otherwise, as many bytes as the workgroup's weights."""
    w64, g64, be64 = w.double(), gamma.double(), beta.double()
    return (w64 * g64[None, :]).float(), (b.double() + w64 @ be64).float()


class RelPosEncoder:
    """espnet-style pre-norm encoder with relative-position attention (text encoder, token encoder,
    and the causal LM body).  Holds one precomputed table ``linear_pos(pe(rel))`` per layer.

    ``fold_ln`` (the LM body): the scale / shift of norm1 / norm2 are folded into the q|k|v and FFN-in projections at
    load (``fold_layernorm``); ``n1`` / ``n2`` / ``after`` then hold ones / zeros (``after_norm`` is folded into the
    output head by the owner).  Prefill and decode use the same folded weights."""

    def __init__(self, sd: SD, prefix: str, heads: int, layers: int, act: str, norm_names: Tuple[str, str],
                 legacy_embed: bool, causal: bool, eps: float, max_pos: int, device, pos_dtype=torch.float32, fold_ln: bool = False):
        self.heads, self.layers, self.act, self.causal, self.eps = heads, layers, act, causal, eps
        self.legacy_embed = legacy_embed
        self.center = max_pos
        self.fold_ln = fold_ln
        p = prefix
        self.d = int(sd[p + ".after_norm.weight"].shape[0])
        self.embed = PackedWeight(sd[p + ".embed.out.0.weight"], sd[p + ".embed.out.0.bias"], device)
        self.embed_ln = (_dev(sd[p + ".embed.out.1.weight"], device), _dev(sd[p + ".embed.out.1.bias"], device))
        ident = (torch.ones(self.d, dtype=torch.float32, device=device), torch.zeros(self.d, dtype=torch.float32, device=device))
        self.after = ident if fold_ln else (_dev(sd[p + ".after_norm.weight"], device), _dev(sd[p + ".after_norm.bias"], device))
        pe = rel_pos_table(self.d, max_pos).to(device)
        n1, n2 = norm_names
        self.L: List[dict] = []
        for i in range(layers):
            q = f"{p}.encoders.{i}"
            a = q + ".self_attn"
            g1, b1 = sd[f"{q}.{n1}.weight"].float().cpu(), sd[f"{q}.{n1}.bias"].float().cpu()
            g2, b2 = sd[f"{q}.{n2}.weight"].float().cpu(), sd[f"{q}.{n2}.bias"].float().cpu()
            wq, bq = sd[a + ".linear_q.weight"].float().cpu(), sd[a + ".linear_q.bias"].float().cpu()
            wkv = torch.cat([sd[a + ".linear_k.weight"], sd[a + ".linear_v.weight"]], 0).float().cpu()
            bkv = torch.cat([sd[a + ".linear_k.bias"], sd[a + ".linear_v.bias"]], 0).float().cpu()
            w1, bw1 = sd[q + ".feed_forward.w_1.weight"].float().cpu(), sd[q + ".feed_forward.w_1.bias"].float().cpu()
            if fold_ln:
                wq, bq = fold_layernorm(wq, bq, g1, b1)
                wkv, bkv = fold_layernorm(wkv, bkv, g1, b1)
                w1, bw1 = fold_layernorm(w1, bw1, g2, b2)
            lay = {
                "n1": ident if fold_ln else (_dev(g1, device), _dev(b1, device)),
                "n2": ident if fold_ln else (_dev(g2, device), _dev(b2, device)),
                "wq": PackedWeight(wq, bq, device),
                "wkv": PackedWeight(wkv, bkv, device),
                "wqkv": PackedWeight(torch.cat([wq, wkv], 0), torch.cat([bq, bkv], 0), device),
                "wo": PackedWeight(sd[a + ".linear_out.weight"], sd[a + ".linear_out.bias"], device),
                "w1": PackedWeight(w1, bw1, device),
                "w2": PackedWeight(sd[q + ".feed_forward.w_2.weight"], sd[q + ".feed_forward.w_2.bias"], device),
                "u": _dev(sd[a + ".pos_bias_u"].reshape(-1), device),
                "v": _dev(sd[a + ".pos_bias_v"].reshape(-1), device),
            }
            lay["pos"] = ops.linear(pe, PackedWeight(sd[a + ".linear_pos.weight"], None, device), out_dtype=pos_dtype)
            self.L.append(lay)

    def embed_in(self, x: torch.Tensor) -> torch.Tensor:
        h = ops.linear(x, self.embed)
        if self.legacy_embed:
            # LayerNorm -> ReLU -> * sqrt(d) in one launch (relu commutes with the positive scale)
            return ops.layernorm(h, *self.embed_ln, self.eps, relu_scale=math.sqrt(self.d))
        h = ops.layernorm(h, *self.embed_ln, self.eps)
        return ops.elementwise(ops.EL_SCALE, h, s=math.sqrt(self.d))

    def forward(self, x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """Batch-major full-sequence pass: x [B, T, in_dim], lens int32 [B] -> [B, T, d]."""
        h = self.embed_in(x)
        d = self.d
        for lay in self.L:
            n = ops.layernorm(h, *lay["n1"], self.eps)
            q = ops.linear(n, lay["wq"])
            kv = ops.linear(n, lay["wkv"])
            a = ops.attn_relpos(q, kv[..., :d], kv[..., d:], lay["pos"], lay["u"], lay["v"], self.heads, lens=lens,
                                q_pos0=0, pos_center=self.center, causal=self.causal)
            h = ops.linear(a, lay["wo"], residual=h)
            n = ops.layernorm(h, *lay["n2"], self.eps)
            f = ops.linear(n, lay["w1"], act=self.act)
            h = ops.linear(f, lay["w2"], residual=h)
        return ops.layernorm(h, *self.after, self.eps)


class AcousticLM:
    """TransformerLM: text encoder + causal rel-pos transformer with a time-major KV cache
    ``[T_max, B, 2d]`` per layer (the K|V GEMM writes straight into the cache rows of the new positions)."""

    def __init__(self, sd: SD, cfg: SynthConfig, device):
        self.cfg, self.device = cfg, device
        self.text_emb = _dev(sd["text_embedding.weight"], device)
        self.text_enc = RelPosEncoder(sd, "text_encoder", cfg.lm_heads, cfg.lm_text_layers, "swish",
                                      ("norm_mha", "norm_ff"), False, True, cfg.ln_eps, cfg.max_positions, device)
        self.text_aff = PackedWeight(sd["text_encoder_affine_layer.weight"], sd["text_encoder_affine_layer.bias"], device)
        self.llm_emb = _dev(sd["llm_embedding.weight"], device)
        self.speech_emb = _dev(sd["speech_embedding.weight"], device)
        self.spk_aff = PackedWeight(sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"], device)
        # the LM body re-reads its position tables and KV cache every decode step: both live in HBM as fp16
        self.body = RelPosEncoder(sd, "llm", cfg.lm_heads, cfg.lm_layers, "relu", ("norm1", "norm2"), True, True,
                                  cfg.ln_eps, cfg.max_positions, device, pos_dtype=torch.float16, fold_ln=True)
        # after_norm is only ever followed by the output head: folded into it (the body's `after` is the identity affine)
        hw, hb = fold_layernorm(sd["llm_decoder.weight"].float().cpu(), sd["llm_decoder.bias"].float().cpu(),
                                sd["llm.after_norm.weight"].float().cpu(), sd["llm.after_norm.bias"].float().cpu())
        self.head = PackedWeight(hw, hb, device)
        import threading
        self._eng_lock = threading.Lock()

    def prefix(self, text: torch.Tensor, text_lens: torch.Tensor, spk: torch.Tensor, prompt_tokens: torch.Tensor) -> torch.Tensor:
        """-> time-major [S0, B, d]: sos | spk | text_encoder(text) | task_id | speech_emb(prompt)."""
        b = text.shape[0]
        te = ops.embedding(self.text_emb, text)
        enc = ops.linear(self.text_enc.forward(te, text_lens), self.text_aff)
        spk_n = torch.nn.functional.normalize(spk, dim=1)  # 192-vector per utterance: host-side plumbing
        spk_e = ops.linear(spk_n, self.spk_aff)[:, None, :]
        sos = self.llm_emb[0].view(1, 1, -1).expand(b, 1, -1)
        task = self.llm_emb[1].view(1, 1, -1).expand(b, 1, -1)
        pe = ops.embedding(self.speech_emb, prompt_tokens)
        return torch.cat([sos, spk_e, enc, task, pe], dim=1).transpose(0, 1).contiguous()

    def new_cache(self, b: int, t_max: int) -> List[torch.Tensor]:
        if t_max > self.body.center:      # relative positions 0 .. t_max - 1 must be rows of the tables
            raise ValueError(f"AcousticLM: a context of {t_max} positions exceeds the relative-position tables "
                             f"(SynthConfig.max_positions = {self.body.center})")
        return [torch.empty((t_max, b, 2 * self.body.d), dtype=torch.float16, device=self.device) for _ in self.body.L]

    def prefix_ragged(self, texts: List[torch.Tensor], spk: torch.Tensor, prompts: List[torch.Tensor]):
        """Ragged batch: row i has its own text length and prompt length.  Rows are LEFT-padded to the longest
        prefix (relative-position attention is translation invariant, so masking the pad keys is exact):
        -> (prefix [S0, B, d] time-major, key_start int32 [B])."""
        b = len(texts)
        dev = self.device
        tl = [int(t.numel()) for t in texts]
        tmax = max(tl)
        text = torch.zeros((b, tmax), dtype=torch.int64, device=dev)
        for i, t in enumerate(texts):
            text[i, :tl[i]] = t.to(dev).view(-1)
        text_lens = torch.tensor(tl, dtype=torch.int32, device=dev)
        te = ops.embedding(self.text_emb, text)
        enc = ops.linear(self.text_enc.forward(te, text_lens), self.text_aff)      # suffix padding + causal: exact
        spk_e = ops.linear(torch.nn.functional.normalize(spk.to(dev), dim=1), self.spk_aff)
        s0 = [2 + tl[i] + 1 + int(prompts[i].numel()) for i in range(b)]
        smax = max(s0)
        pre = torch.zeros((b, smax, self.body.d), dtype=torch.float32, device=dev)
        for i in range(b):                                                          # host-side assembly (plumbing)
            o = smax - s0[i]
            pre[i, o] = self.llm_emb[0]
            pre[i, o + 1] = spk_e[i]
            pre[i, o + 2:o + 2 + tl[i]] = enc[i, :tl[i]]
            pre[i, o + 2 + tl[i]] = self.llm_emb[1]
            pre[i, o + 3 + tl[i]:] = ops.embedding(self.speech_emb, prompts[i].to(dev).view(1, -1))[0]
        key_start = torch.tensor([smax - s for s in s0], dtype=torch.int32, device=dev)
        return pre.transpose(0, 1).contiguous(), key_start

    def forward_new(self, x: torch.Tensor, cache: List[torch.Tensor], pos0: int, key_start: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: time-major [T, B, d] NEW positions pos0..pos0+T-1 -> hidden [T, B, d]; fills the cache."""
        body = self.body
        d = body.d
        t, b = x.shape[0], x.shape[1]
        h = body.embed_in(x)
        tk = pos0 + t
        lens = torch.full((b,), tk, dtype=torch.int32, device=self.device)
        for lay, kvc in zip(body.L, cache):
            n = ops.layernorm(h, *lay["n1"], body.eps)
            q = ops.linear(n, lay["wq"])
            ops.gemm(n.view(t * b, d), lay["wkv"], out=kvc[pos0:pos0 + t].view(t * b, 2 * d))
            kv = kvc[:tk]
            a = ops.attn_relpos(q, kv[..., :d], kv[..., d:], lay["pos"], lay["u"], lay["v"], body.heads, lens=lens,
                                q_pos0=pos0, pos_center=body.center, causal=True, time_major=True, key_start=key_start)
            h = ops.linear(a, lay["wo"], residual=h)
            n = ops.layernorm(h, *lay["n2"], body.eps)
            f = ops.linear(n, lay["w1"], act="relu")
            h = ops.linear(f, lay["w2"], residual=h)
        return ops.layernorm(h, *body.after, body.eps)

    def logits(self, hidden_last: torch.Tensor) -> torch.Tensor:
        return ops.linear(hidden_last, self.head)

    def step_logits(self, tok: torch.Tensor, cache: List[torch.Tensor], pos: int, key_start: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One decode step for B <= 32 utterances with the launch-saving fusions of astts_op_gemm_fused:
        embedding gather inside the embed GEMM, LayerNorm inside the QKV / FFN-in / head GEMMs, K|V written
        straight into the cache.  5 launches per layer.  tok int32 [B] -> logits [B, V+1]."""
        body = self.body
        d, b = body.d, tok.shape[0]
        h = ops.gemm_fused(self.speech_emb, body.embed, b, gather=tok)
        h = ops.layernorm(h, *body.embed_ln, body.eps, relu_scale=math.sqrt(d))
        lens = torch.full((b,), pos + 1, dtype=torch.int32, device=self.device)
        for lay, kvc in zip(body.L, cache):
            q = ops.gemm_fused(h, lay["wqkv"], b, ln=lay["n1"], ln_eps=body.eps, out2=kvc[pos], n_split=d)
            kv = kvc[:pos + 1]
            a = ops.attn_relpos(q[None], kv[..., :d], kv[..., d:], lay["pos"], lay["u"], lay["v"], body.heads, lens=lens,
                                q_pos0=pos, pos_center=body.center, causal=True, time_major=True, key_start=key_start)
            h = ops.gemm_fused(a[0], lay["wo"], b, residual=h)
            f = ops.gemm_fused(h, lay["w1"], b, ln=lay["n2"], ln_eps=body.eps, act="relu")
            h = ops.gemm_fused(f, lay["w2"], b, residual=h)
        return ops.gemm_fused(h, self.head, b, ln=body.after, ln_eps=body.eps)

    # ---- C++ decode engine (libastts astts_lm_*): same fused step, issued without Python in the loop
    def _engine(self):
        if getattr(self, "_eng", None) is not None:
            return self._eng
        with self._eng_lock:            # decode groups run on several host threads: ONE of them builds the handle and its table
            return self._engine_locked()

    def _engine_locked(self):
        if getattr(self, "_eng", None) is None:
            import ctypes

            from .. import _lib
            body, cfg = self.body, self.cfg
            c = ops.LmConfig(body.d, body.heads, cfg.lm_ffn, len(body.L), cfg.speech_vocab + 1, cfg.speech_vocab, body.center,
                             body.L[0]["pos"].stride(0), cfg.top_k, cfg.ras_win, cfg.top_p, cfg.ras_tau, body.eps, 1, 1,
                             1 if body.fold_ln else 0, 1 if cfg.eos_policy == "reject" else 0)
            # the decode step's input projection acts on a table lookup: speech_emb[tok] W^T + b is a row of the table
            # speech_emb W^T + b, formed once here (the same fp16 products on the GPU; the big-tile GEMM sums them in another fp32
            # order than the step's GEMV, so the two paths agree to rounding, not bit for bit: tests/test_lm_step_gpu.py) and
            # gathered by the step.  ASTTS_LM_EMBED_TABLE=0: no table, the projection runs every step (the engine's other branch).
            import os
            self.embed_table = ops.linear(self.speech_emb, body.embed).contiguous() if os.environ.get("ASTTS_LM_EMBED_TABLE", "1") != "0" else None
            torch.cuda.current_stream(self.device).synchronize()     # one-off: decode chains on OTHER streams read the table
            g = ops.LmGlobals(self.speech_emb.data_ptr(), body.embed.data.data_ptr(), body.embed.bias.data_ptr(),
                              body.embed_ln[0].data_ptr(), body.embed_ln[1].data_ptr(), body.after[0].data_ptr(),
                              body.after[1].data_ptr(), self.head.data.data_ptr(), self.head.bias.data_ptr(),
                              None if self.embed_table is None else self.embed_table.data_ptr())
            arr = (ops.LmLayer * len(body.L))()
            for i, L in enumerate(body.L):
                arr[i] = ops.LmLayer(L["n1"][0].data_ptr(), L["n1"][1].data_ptr(), L["wqkv"].data.data_ptr(), L["wqkv"].bias.data_ptr(),
                                     L["wo"].data.data_ptr(), L["wo"].bias.data_ptr(), L["n2"][0].data_ptr(), L["n2"][1].data_ptr(),
                                     L["w1"].data.data_ptr(), L["w1"].bias.data_ptr(), L["w2"].data.data_ptr(), L["w2"].bias.data_ptr(),
                                     L["pos"].data_ptr(), L["u"].data_ptr(), L["v"].data_ptr())
            h = ctypes.c_void_p()
            _lib.check(_lib.load().astts_lm_create(ctypes.byref(c), ctypes.byref(g), arr, ctypes.byref(h)))
            self._eng = h
        return self._eng

    @staticmethod
    def _eos_min(ignore_eos, n_steps: int) -> int:
        """True -> EOS masked for the whole fixed-length decode; False -> never; int -> masked for that many steps;
        int32 device tensor [B] -> per-row step counts (handled by the engine; scalar fallback 0)."""
        if torch.is_tensor(ignore_eos):
            return 0
        if ignore_eos is True:
            return n_steps
        if ignore_eos is False:
            return 0
        return int(ignore_eos)

    def prefill(self, prefix: torch.Tensor, n_steps: int, key_start: Optional[torch.Tensor] = None):
        """The part of a decode call that does not depend on the sampler: the KV cache of the prefix and the logits of its last
        position, on the CURRENT stream.  -> state for `decode_prefilled` (which may run on another stream: the pipeline enqueues
        this on its front stream so that a decode chain is nothing but its steps)."""
        s0, b = prefix.shape[0], prefix.shape[1]
        cache = self.new_cache(b, s0 + n_steps)
        hid = self.forward_new(prefix, cache, 0, key_start)
        logits0 = self.logits(hid[-1]).contiguous()
        return {"cache": cache, "logits0": logits0, "s0": s0, "b": b, "n_steps": n_steps, "key_start": key_start}

    @staticmethod
    def prefill_tensors(state):
        return list(state["cache"]) + [state["logits0"]] + ([state["key_start"]] if state["key_start"] is not None else [])

    def decode_begin(self, state, uniforms: torch.Tensor, ignore_eos=True, forced_tokens: Optional[torch.Tensor] = None,
                     return_logits: bool = False):
        """The buffers of one decode of `prefill`'s state (b <= 32 rows): token history, workspace (it carries the logits from one
        range of steps to the next), uniforms.  -> context for `decode_range`."""
        from .. import _lib
        lib = _lib.load()
        eng = self._engine()
        b, n_steps = state["b"], state["n_steps"]
        need = int(lib.astts_lm_workspace_bytes(eng, b))
        ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        ctx = dict(state)
        ctx.update({"ws": ws, "ws_aligned": (ws.data_ptr() + 255) // 256 * 256, "ws_bytes": need,
                    "toks": torch.zeros((b, n_steps), dtype=torch.int32, device=self.device),
                    "logits": torch.empty((b, n_steps, self.cfg.speech_vocab + 1), dtype=torch.float32, device=self.device) if return_logits else None,
                    "forced": None if forced_tokens is None else forced_tokens.to(torch.int32).contiguous(),
                    "u": uniforms.to(torch.float32).contiguous(), "ignore_eos": ignore_eos, "next": 0})
        return ctx

    def decode_range(self, ctx, s_end: Optional[int] = None) -> None:
        """Enqueue decode steps [ctx["next"], s_end) on the CURRENT stream (one C++ call, no host synchronisation).  The ranges of
        one decode must be issued in order on one stream (astts_lm_decode_range)."""
        import ctypes

        from .. import _lib
        lib = _lib.load()
        n_steps, s0, b, cache, key_start, ignore_eos = (ctx[k] for k in ("n_steps", "s0", "b", "cache", "key_start", "ignore_eos"))
        s_begin = ctx["next"]
        s_end = n_steps if s_end is None else min(int(s_end), n_steps)
        if s_end <= s_begin:
            return
        ptrs = (ctypes.c_void_p * len(cache))(*[c.data_ptr() for c in cache])
        _lib.check(lib.astts_lm_decode_range(self._engine(), ctx["logits0"].data_ptr(), ptrs, None if key_start is None else key_start.data_ptr(),
                                             s0 + n_steps, b, s0, n_steps, s_begin, s_end, ctx["u"].data_ptr(),
                                             None if ctx["forced"] is None else ctx["forced"].data_ptr(), self._eos_min(ignore_eos, n_steps),
                                             ignore_eos.data_ptr() if torch.is_tensor(ignore_eos) else None, ctx["toks"].data_ptr(),
                                             None if ctx["logits"] is None else ctx["logits"].data_ptr(), ctx["ws_aligned"], ctx["ws_bytes"],
                                             _lib.stream_ptr()))
        ctx["next"] = s_end

    def decode_prefilled(self, state, uniforms: torch.Tensor, ignore_eos: bool = True, forced_tokens: Optional[torch.Tensor] = None,
                         return_logits: bool = False):
        """The decode steps of `prefill`'s state (b <= 32 rows) on the current stream: one C++ call, no host synchronisation."""
        ctx = self.decode_begin(state, uniforms, ignore_eos, forced_tokens, return_logits)
        self.decode_range(ctx)
        self._keepalive = ctx            # buffers referenced by kernels still in flight
        return (ctx["toks"], ctx["logits"]) if return_logits else ctx["toks"]

    def decode_engine(self, prefix: torch.Tensor, n_steps: int, uniforms: torch.Tensor, ignore_eos: bool = True,
                      forced_tokens: Optional[torch.Tensor] = None, return_logits: bool = False,
                      key_start: Optional[torch.Tensor] = None):
        return self.decode_prefilled(self.prefill(prefix, n_steps, key_start), uniforms, ignore_eos, forced_tokens, return_logits)

    WIDE_ROWS = 256      # ASTTS_LM_MAX_ROWS: rows of one wide-engine call

    def decode(self, prefix: torch.Tensor, n_steps: int, uniforms: torch.Tensor, ignore_eos: bool = True,
               forced_tokens: Optional[torch.Tensor] = None, return_logits: bool = False, use_engine: bool = True,
               key_start: Optional[torch.Tensor] = None, group_steps: Optional[List[int]] = None, wide: bool = False):
        """Fixed-length autoregressive decode, no host synchronisation inside the loop.
        prefix: [S0, B, d]; uniforms [n_steps, B, 2] -> tokens int32 [B, n_steps] (+ logits [B, n_steps, V+1]).
        ``wide`` (throughput runs): batches of 33 .. 256 rows go through ONE chain of plain GEMMs per projection (the engine's wide
        path: the weights are read once per token for all rows) instead of 32-row groups on two streams.  Same fp16 products with fp32
        accumulation, another summation order: a row's tokens then need not equal its <= 32-row run bit for bit (the default keeps the
        groups, whose rows do)."""
        cfg = self.cfg
        s0, b = prefix.shape[0], prefix.shape[1]
        if use_engine and (b <= 32 or (wide and b <= self.WIDE_ROWS)):
            return self.decode_engine(prefix, n_steps, uniforms, ignore_eos, forced_tokens, return_logits, key_start)
        if use_engine:
            # larger batches (BASELINE config 3: 64 long-form utterances): independent rows, decoded in groups of 32.  The
            # groups are separate launch chains, latency-bound like any decode, so two of them run concurrently on their
            # own streams (each enqueued by its own host thread: the engine call drops the GIL); results do not depend on
            # the grouping or on what runs beside them.
            import threading

            groups = [slice(b0, min(b0 + 32, b)) for b0 in range(0, b, 32)]
            # ``group_steps``: group g decodes only group_steps[g] <= n_steps steps (ragged batches sorted by length: the rows of a
            # later group need fewer); its tokens come back zero-padded to n_steps
            if group_steps is not None and (len(group_steps) != len(groups) or return_logits or max(group_steps) > n_steps):
                raise ValueError("AcousticLM.decode: group_steps needs one step count <= n_steps per 32-row group (and no return_logits)")
            if getattr(self, "_group_streams", None) is None:
                self._group_streams = ops.concurrent_streams(2, device=self.device)
            cur = torch.cuda.current_stream(self.device)
            outs, errs = [None] * len(groups), []

            def run(gi, st):
                try:
                    sl = groups[gi]
                    with torch.cuda.device(self.device), torch.cuda.stream(st):
                        n_g = n_steps if group_steps is None else int(group_steps[gi])
                        o = self.decode_engine(prefix[:, sl].contiguous(), n_g, uniforms[:n_g, sl].contiguous(),
                                               ignore_eos[sl].contiguous() if torch.is_tensor(ignore_eos) else ignore_eos,
                                               None if forced_tokens is None else forced_tokens[sl, :n_g].contiguous(), return_logits,
                                               None if key_start is None else key_start[sl].contiguous())
                        if n_g < n_steps:
                            o = torch.nn.functional.pad(o, (0, n_steps - n_g))
                        outs[gi] = o
                except BaseException as e:      # noqa: BLE001  (re-raised on the calling thread)
                    errs.append(e)

            for w0 in range(0, len(groups), 2):
                wave = []
                for k, gi in enumerate(range(w0, min(w0 + 2, len(groups)))):
                    st = self._group_streams[k]
                    st.wait_stream(cur)
                    th = threading.Thread(target=run, args=(gi, st))
                    th.start()
                    wave.append((th, st))
                for th, st in wave:
                    th.join()
                    cur.wait_stream(st)
            if errs:
                raise errs[0]
            for o in outs:
                for t_ in (o if isinstance(o, tuple) else (o,)):
                    t_.record_stream(cur)
            if return_logits:
                return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)
            return torch.cat(outs, 0)
        cache = self.new_cache(b, s0 + n_steps)
        hid = self.forward_new(prefix, cache, 0, key_start)
        cur = self.logits(hid[-1])
        toks = torch.zeros((b, n_steps), dtype=torch.int32, device=self.device)
        all_logits = [] if return_logits else None
        for s in range(n_steps):
            if return_logits:
                all_logits.append(cur)
            tok = ops.ras_sample(cur, toks, s, uniforms[s], cfg.top_k, cfg.top_p, cfg.ras_win, cfg.ras_tau,
                                 cfg.speech_vocab, s < self._eos_min(ignore_eos, n_steps), eos_policy=cfg.eos_policy)
            if forced_tokens is not None:
                tok = forced_tokens[:, s].to(torch.int32).contiguous()
            toks[:, s] = tok
            if s + 1 < n_steps:
                if b <= 32:
                    cur = self.step_logits(tok.clamp(max=cfg.speech_vocab - 1), cache, s0 + s, key_start)
                else:
                    emb = ops.embedding(self.speech_emb, tok)[None]        # [1, B, d]
                    hid = self.forward_new(emb, cache, s0 + s, key_start)
                    cur = self.logits(hid[0])
        if return_logits:
            return toks, torch.stack(all_logits, dim=1)
        return toks


class _Resnet1D:
    def __init__(self, sd: SD, p: str, device, groups: int):
        self.groups = groups
        self.c1 = PackedWeight.from_conv1d(sd[p + ".block1.block.0.weight"], sd[p + ".block1.block.0.bias"], device)
        self.g1 = (_dev(sd[p + ".block1.block.1.weight"], device), _dev(sd[p + ".block1.block.1.bias"], device))
        self.mlp = PackedWeight(sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"], device)
        self.c2 = PackedWeight.from_conv1d(sd[p + ".block2.block.0.weight"], sd[p + ".block2.block.0.bias"], device)
        self.g2 = (_dev(sd[p + ".block2.block.1.weight"], device), _dev(sd[p + ".block2.block.1.bias"], device))
        self.res = PackedWeight.from_conv1d(sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"], device)
        # 256 -> 256 blocks (all mid blocks): three launches with the GroupNorm + Mish passes folded into the convolutions
        # (ops.resnet_conv) instead of five
        self.fused = all(w.cin == w.cin_pad for w in (self.c1, self.c2, self.res)) and \
            ops.resnet_conv_supported(self.c1.cin, self.c1.n, groups, self.c1.taps) and \
            ops.resnet_conv_supported(self.c2.cin, self.c2.n, groups, self.c2.taps) and \
            ops.resnet_conv_supported(self.res.cin, self.res.n, groups, self.res.taps)
        self.c1_frag, self.c2_frag, self.res_frag = ((ops.conv_pack_frag(self.c1), ops.conv_pack_frag(self.c2), ops.conv_pack_frag(self.res))
                                                     if self.fused else (None, None, None))

    def forward(self, x: torch.Tensor, lens: torch.Tensor, temb_mish: torch.Tensor, tproj: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x must already be zero beyond lens (masked).  ``tproj`` [B, C]: this block's time projection when the caller has
        it already (the solver evaluates the time path of all Euler steps up front)."""
        if tproj is None:
            tproj = ops.linear(temb_mish, self.mlp)                              # [B, C]
        if self.fused:
            x = x.contiguous()
            h1, s1 = ops.resnet_conv(x, self.c1, self.c1_frag, lens=lens, want_stats=True)
            h2, s2 = ops.resnet_conv(h1, self.c2, self.c2_frag, lens=lens, in_gn=(s1, *self.g1), in_add=tproj.contiguous(), want_stats=True)
            return ops.resnet_conv(x, self.res, self.res_frag, lens=lens, res_gn=(h2, s2, *self.g2))
        h = ops.conv1d(x, self.c1, pad=1)
        h = ops.groupnorm(h, *self.g1, self.groups, 1e-5, lens=lens, mish=True, add_bc=tproj, out_dtype=torch.float16)
        h = ops.conv1d(h, self.c2, pad=1)                                       # fp16 in: only consumer is the MFMA
        h = ops.groupnorm(h, *self.g2, self.groups, 1e-5, lens=lens, mish=True)
        return ops.conv1d(x, self.res, residual=h)                              # res_conv(x) + h


class _TfmBlock:
    def __init__(self, sd: SD, p: str, heads: int, device):
        self.heads = heads
        # norm1's scale / shift are folded into the fused q|k|v projection (W' = W diag(gamma), b' = W beta: fold_layernorm), so
        # that the fused LayerNorm + projection + attention kernel (ops.tfm_attn_fused) normalises without them; the unfused
        # path (T > 384) runs the plain LayerNorm kernel with the identity affine on the same weights.
        wqkv = torch.cat([sd[p + ".attn1.to_q.weight"], sd[p + ".attn1.to_k.weight"], sd[p + ".attn1.to_v.weight"]], 0).float().cpu()
        wqkv, bqkv = fold_layernorm(wqkv, torch.zeros(wqkv.shape[0]), sd[p + ".norm1.weight"].float().cpu(), sd[p + ".norm1.bias"].float().cpu())
        c = int(wqkv.shape[1])
        self.n1 = (torch.ones(c, dtype=torch.float32, device=device), torch.zeros(c, dtype=torch.float32, device=device))
        self.wqkv = PackedWeight(wqkv, bqkv, device)
        self.wqkv_frag = ops.tfm_pack_frag(self.wqkv) if c == 256 and wqkv.shape[0] % 32 == 0 else None   # fused kernel's weight image
        self.wo = PackedWeight(sd[p + ".attn1.to_out.0.weight"], sd[p + ".attn1.to_out.0.bias"], device)
        # norm3 folded into the first feed-forward projection in the same way (ops.tfm_ffn_fused: LayerNorm + W1 + GELU + W2 +
        # residual in one launch)
        w1, b1 = fold_layernorm(sd[p + ".ff.net.0.proj.weight"].float().cpu(), sd[p + ".ff.net.0.proj.bias"].float().cpu(),
                                sd[p + ".norm3.weight"].float().cpu(), sd[p + ".norm3.bias"].float().cpu())
        self.n3 = self.n1
        self.w1 = PackedWeight(w1, b1, device)
        self.w2 = PackedWeight(sd[p + ".ff.net.2.weight"], sd[p + ".ff.net.2.bias"], device)
        self.ffn_fused = ops.tfm_ffn_fused_supported(c, int(w1.shape[0]))
        self.w1_frag = ops.tfm_pack_frag(self.w1) if self.ffn_fused else None
        self.w2_frag = ops.tfm_pack_frag(self.w2) if self.ffn_fused else None
        # ... and the attention's output projection + residual as that launch's prologue
        self.wo_frag = ops.tfm_pack_frag(self.wo) if self.ffn_fused and self.wo.cin in (256, 512) else None

    def forward(self, x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        hd = self.heads * 64
        f16 = torch.float16   # everything between two residual adds feeds MFMA operands only: fp16 in HBM
        if self.wqkv_frag is not None and ops.tfm_attn_fused_supported(x.shape[-1], self.heads, x.shape[1]):
            a = ops.tfm_attn_fused(x, self.wqkv, self.wqkv_frag, self.heads, lens=lens, eps=1e-5)     # LayerNorm + q|k|v + attention: one launch
        else:
            n = ops.layernorm(x, *self.n1, 1e-5, out_dtype=f16)
            qkv = ops.linear(n, self.wqkv, out_dtype=f16)
            a = ops.attn_mha(qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:], self.heads, lens=lens, out_dtype=f16)
        if self.wo_frag is not None:
            return ops.tfm_ffn_fused(x, self.w1, self.w1_frag, self.w2, self.w2_frag, eps=1e-5, attn=a.contiguous(), wo=self.wo, wo_frag=self.wo_frag)
        x = ops.linear(a, self.wo, residual=x)
        if self.ffn_fused:
            return ops.tfm_ffn_fused(x, self.w1, self.w1_frag, self.w2, self.w2_frag, eps=1e-5)
        n = ops.layernorm(x, *self.n3, 1e-5, out_dtype=f16)
        f = ops.linear(n, self.w1, act="gelu", out_dtype=f16)
        return ops.linear(f, self.w2, residual=x)


class FlowDecoder:
    """MaskedDiffWithXvec + ConditionalCFM: tokens -> mu -> 10 Euler steps of the U-Net estimator with
    classifier-free guidance (cond and uncond halves run as one 2B batch)."""

    def __init__(self, sd: SD, cfg: SynthConfig, device):
        self.cfg, self.device = cfg, device
        self.use_engine = True          # False: operator-by-operator solve from Python (tests compare the two)
        self.tok_emb = _dev(sd["input_embedding.weight"], device)
        self.spk_aff = PackedWeight(sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"], device)
        self.enc = RelPosEncoder(sd, "encoder", cfg.flow_heads, cfg.flow_layers, "swish", ("norm_mha", "norm_ff"), False,
                                 False, cfg.ln_eps, cfg.max_positions, device)
        self.enc_proj = PackedWeight(sd["encoder_proj.weight"], sd["encoder_proj.bias"], device)
        self.lr = []
        for j in range(4):
            self.lr.append((PackedWeight.from_conv1d(sd[f"length_regulator.model.{3 * j}.weight"],
                                                     sd[f"length_regulator.model.{3 * j}.bias"], device),
                            _dev(sd[f"length_regulator.model.{3 * j + 1}.weight"], device),
                            _dev(sd[f"length_regulator.model.{3 * j + 1}.bias"], device)))
        self.lr_out = PackedWeight.from_conv1d(sd["length_regulator.model.12.weight"], sd["length_regulator.model.12.bias"], device)
        e = "decoder.estimator"
        self.t1 = PackedWeight(sd[e + ".time_mlp.linear_1.weight"], sd[e + ".time_mlp.linear_1.bias"], device)
        self.t2 = PackedWeight(sd[e + ".time_mlp.linear_2.weight"], sd[e + ".time_mlp.linear_2.bias"], device)
        ch = cfg.est_channels
        g = cfg.est_groups
        self.down, self.mid, self.up = [], [], []
        for i in range(len(ch)):
            p = f"{e}.down_blocks.{i}"
            last = i == len(ch) - 1
            wname = p + ".2" + ("" if last else ".conv")
            self.down.append((_Resnet1D(sd, p + ".0", device, g),
                              [_TfmBlock(sd, f"{p}.1.{j}", cfg.est_heads, device) for j in range(cfg.est_tfm_per_block)],
                              PackedWeight.from_conv1d(sd[wname + ".weight"], sd[wname + ".bias"], device), last))
        for i in range(cfg.est_mid_blocks):
            p = f"{e}.mid_blocks.{i}"
            self.mid.append((_Resnet1D(sd, p + ".0", device, g),
                             [_TfmBlock(sd, f"{p}.1.{j}", cfg.est_heads, device) for j in range(cfg.est_tfm_per_block)]))
        for i in range(len(ch)):
            p = f"{e}.up_blocks.{i}"
            last = i == len(ch) - 1
            if last:
                w = PackedWeight.from_conv1d(sd[p + ".2.weight"], sd[p + ".2.bias"], device)
            else:
                w = PackedWeight.from_conv_transpose1d(sd[p + ".2.conv.weight"], sd[p + ".2.conv.bias"], 2, device)
            self.up.append((_Resnet1D(sd, p + ".0", device, g),
                            [_TfmBlock(sd, f"{p}.1.{j}", cfg.est_heads, device) for j in range(cfg.est_tfm_per_block)], w, last))
        self.fin_c = PackedWeight.from_conv1d(sd[e + ".final_block.block.0.weight"], sd[e + ".final_block.block.0.bias"], device)
        self.fin_g = (_dev(sd[e + ".final_block.block.1.weight"], device), _dev(sd[e + ".final_block.block.1.bias"], device))
        self.fin_p = PackedWeight.from_conv1d(sd[e + ".final_proj.weight"], sd[e + ".final_proj.bias"], device)

    # ---- token encoder + length regulator
    def mu(self, tokens: torch.Tensor, token_lens: torch.Tensor, mel_total: int, mel_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mel_lens (ragged batch): row b is regulated from token_lens[b] tokens to mel_lens[b] frames, zeros beyond."""
        x = ops.embedding(self.tok_emb, tokens.clamp(min=0))
        x = ops.elementwise(ops.EL_MUL_ROWMASK, x, lens=token_lens)
        h = ops.linear(self.enc.forward(x, token_lens), self.enc_proj)
        h = ops.interp_linear(h, mel_total, in_lens=None if mel_lens is None else token_lens, out_lens=mel_lens)
        for w, ga, be in self.lr:
            h = ops.conv1d(h, w, pad=1)
            h = ops.groupnorm(h, ga, be, 1, 1e-5, lens=mel_lens, mish=True)      # writes 0 beyond the row's length
        h = ops.conv1d(h, self.lr_out)
        return h if mel_lens is None else ops.elementwise(ops.EL_MUL_ROWMASK, h, lens=mel_lens)

    # ---- one estimator evaluation on a (2B) batch
    def time_path(self, t: torch.Tensor) -> List[torch.Tensor]:
        """t fp32 [rows] -> the time projection of every ResnetBlock1D (down, mid, up order), each [rows, C]: sinusoidal
        embedding -> MLP -> Mish (every block applies Mish before its own Linear).  Independent of x: the solver calls it once
        with the rows of ALL Euler steps."""
        temb = ops.time_embedding(t, self.cfg.est_in)
        temb = ops.linear(ops.linear(temb, self.t1, act="silu"), self.t2)
        temb_m = ops.elementwise(ops.EL_MISH, temb)
        return [ops.linear(temb_m, blk[0].mlp) for grp in (self.down, self.mid, self.up) for blk in grp]

    def estimator(self, x, mu, spk, cond, t, lens, full: bool = False, tprojs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """``full``: every row uses all T frames (fixed-length batch) -- the length masks are identities and are
        not launched.  ``tprojs``: the blocks' time projections for this call's rows (``time_path``), computed here if absent."""
        cfg = self.cfg
        b, T, _ = x.shape
        if full:
            _mask = lambda h, L: h
        else:
            _mask = lambda h, L: ops.elementwise(ops.EL_MUL_ROWMASK, h, lens=L)
        if tprojs is None:
            tprojs = self.time_path(t)
        tp = iter(tprojs)
        temb_m = None
        h = torch.cat([x, mu, spk[:, None, :].expand(b, T, -1), cond], dim=-1)
        hiddens, lens_stack = [], [lens]
        for res, tfms, wds, last in self.down:
            L = lens_stack[-1]
            h = _mask(h, L)
            h = res.forward(h, L, temb_m, next(tp))
            for tb in tfms:
                h = tb.forward(h, L)
            hiddens.append(h)
            hm = _mask(h, L)
            if last:
                h = ops.conv1d(hm, wds, pad=1)
                lens_stack.append(L)
            else:
                h = ops.conv1d(hm, wds, stride=2, pad=1)
                lens_stack.append(torch.div(L + 1, 2, rounding_mode="floor").to(torch.int32))
        L = lens_stack[-1]
        h = _mask(h, L)
        for res, tfms in self.mid:
            h = res.forward(h, L, temb_m, next(tp))
            for tb in tfms:
                h = tb.forward(h, L)
            h = _mask(h, L)
        lens_stack.pop()
        for res, tfms, wus, last in self.up:
            L = lens_stack.pop()
            skip = hiddens.pop()
            h = torch.cat([h[:, :skip.shape[1]], skip], dim=-1)
            h = _mask(h, L)
            h = res.forward(h, L, temb_m, next(tp))
            for tb in tfms:
                h = tb.forward(h, L)
            hm = _mask(h, L)
            if last:
                h = ops.conv1d(hm, wus, pad=1)
            else:
                h = ops.conv_transpose1d(hm, wus, padding=1)
        h = _mask(h[:, :T].contiguous(), lens)
        h = ops.conv1d(h, self.fin_c, pad=1)
        h = ops.groupnorm(h, *self.fin_g, cfg.est_groups, 1e-5, lens=lens, mish=True)
        out = ops.conv1d(h, self.fin_p)
        return _mask(out, lens)

    # ---- C++ solver engine (libastts astts_flow_*): the same operator sequence, issued without Python in the loop
    def _engine(self):
        if getattr(self, "_eng", None) is None:
            import ctypes

            from .. import _lib
            cfg = self.cfg
            keep = []           # ctypes arrays referenced by pointer from the block structs

            def W(pw):
                return ops.Weight(pw.data.data_ptr(), None if pw.bias is None else pw.bias.data_ptr(), pw.n, pw.cin, pw.cin_pad, pw.taps)

            def R(r):
                fp = [None if f is None else f.data_ptr() for f in (r.c1_frag, r.c2_frag, r.res_frag)]
                return ops.FlowResnet(W(r.c1), W(r.mlp), W(r.c2), W(r.res), r.g1[0].data_ptr(), r.g1[1].data_ptr(),
                                      r.g2[0].data_ptr(), r.g2[1].data_ptr(), *fp)

            def TF(tfms):
                arr = (ops.FlowTfm * len(tfms))()
                for i, t in enumerate(tfms):
                    arr[i] = ops.FlowTfm(t.n1[0].data_ptr(), t.n1[1].data_ptr(), t.n3[0].data_ptr(), t.n3[1].data_ptr(),
                                         W(t.wqkv), W(t.wo), W(t.w1), W(t.w2), None if t.wqkv_frag is None else t.wqkv_frag.data_ptr(),
                                         None if t.w1_frag is None else t.w1_frag.data_ptr(), None if t.w2_frag is None else t.w2_frag.data_ptr(),
                                         None if t.wo_frag is None else t.wo_frag.data_ptr())
                keep.append(arr)
                return arr

            def blocks(items, kinds):
                arr = (ops.FlowBlock * max(len(items), 1))()
                for i, (it, kind) in enumerate(zip(items, kinds)):
                    rs = W(it[2]) if kind != ops.FLOW_RESAMPLE_NONE else ops.Weight()
                    arr[i] = ops.FlowBlock(R(it[0]), TF(it[1]), len(it[1]), rs, kind)
                keep.append(arr)
                return arr

            down = blocks(self.down, [ops.FLOW_RESAMPLE_CONV if it[3] else ops.FLOW_RESAMPLE_DOWN for it in self.down])
            mid = blocks(self.mid, [ops.FLOW_RESAMPLE_NONE] * len(self.mid))
            up = blocks(self.up, [ops.FLOW_RESAMPLE_CONV if it[3] else ops.FLOW_RESAMPLE_UP for it in self.up])
            c = ops.FlowConfig(cfg.mel, cfg.est_channels[0], cfg.est_heads, cfg.est_groups, cfg.est_in, cfg.est_time_dim,
                               len(self.down), len(self.mid), len(self.up), W(self.t1), W(self.t2), W(self.fin_c), W(self.fin_p),
                               self.fin_g[0].data_ptr(), self.fin_g[1].data_ptr())
            h = ctypes.c_void_p()
            _lib.check(_lib.load().astts_flow_create(ctypes.byref(c), down, mid, up, ctypes.byref(h)))
            self._eng, self._eng_keep = h, keep
        return self._eng

    def _schedule(self):
        n = self.cfg.cfm_steps
        ts = 1.0 - torch.cos(torch.linspace(0, 1, n + 1) * 0.5 * math.pi)       # cosine schedule (fp32, as upstream)
        return [float(ts[s]) for s in range(n)], [float(ts[s + 1] - ts[s]) for s in range(n)]

    def solve(self, x: torch.Tensor, mu: torch.Tensor, spk_e: torch.Tensor, cond: torch.Tensor,
              lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Euler solve with classifier-free guidance, in place on ``x`` [B, T, mel] (in: noise, out: mel).
        ``lens`` None = fixed-length batch.  Runs in the C++ engine; ``solve_ops`` is the operator-by-operator form."""
        import ctypes

        from .. import _lib
        lib = _lib.load()
        eng = self._engine()
        b, t, _ = x.shape
        assert x.is_contiguous() and mu.is_contiguous() and cond.is_contiguous() and spk_e.is_contiguous()
        assert x.dtype == mu.dtype == cond.dtype == spk_e.dtype == torch.float32
        tv, dv = self._schedule()
        n = len(tv)
        need = int(lib.astts_flow_workspace_bytes(eng, b, t))
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _lib.check(lib.astts_flow_solve(eng, x.data_ptr(), mu.data_ptr(), spk_e.data_ptr(), cond.data_ptr(),
                                        None if lens is None else lens.data_ptr(), b, t, n, (ctypes.c_float * n)(*tv),
                                        (ctypes.c_float * n)(*dv), self.cfg.cfg_rate, ws.data_ptr(), need,
                                        _lib.stream_ptr()))
        return x

    def solve_ops(self, x, mu, spk_e, cond, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same solve, one Python call per operator (the form the C++ engine is checked against, bit for bit)."""
        b, t, _ = x.shape
        full = lens is None
        if full:
            lens = torch.full((b,), t, dtype=torch.int32, device=x.device)
        lens2 = torch.cat([lens, lens])
        mu2 = torch.cat([mu, torch.zeros_like(mu)], 0)
        spk2 = torch.cat([spk_e, torch.zeros_like(spk_e)], 0)
        cond2 = torch.cat([cond, torch.zeros_like(cond)], 0)
        tv, dv = self._schedule()
        b2 = 2 * b
        t_all = torch.tensor(tv, dtype=torch.float32, device=x.device).repeat_interleave(b2)      # rows of all steps, as the engine
        tp_all = self.time_path(t_all)
        for s in range(len(tv)):
            t2 = t_all[s * b2:(s + 1) * b2]
            d = self.estimator(torch.cat([x, x], 0), mu2, spk2, cond2, t2, lens2, full=full,
                               tprojs=[tp[s * b2:(s + 1) * b2] for tp in tp_all])
            x = ops.elementwise(ops.EL_CFG_EULER, x, z=d, s=dv[s], s2=self.cfg.cfg_rate)
        return x

    def decode(self, tokens, token_lens, prompt_mel, spk, z, mel_total: int) -> torch.Tensor:
        """tokens [B, Tp+Ts], prompt_mel [B, Tm_p, mel], spk [B, spk_dim], z [B, mel_total, mel]
        -> mel [B, mel_total - Tm_p, mel] (fixed-length batch)."""
        cfg = self.cfg
        b = tokens.shape[0]
        mu = self.mu(tokens, token_lens, mel_total)
        spk_e = ops.linear(torch.nn.functional.normalize(spk, dim=1), self.spk_aff)
        tmp = prompt_mel.shape[1]
        cond = torch.zeros((b, mel_total, cfg.mel), dtype=torch.float32, device=self.device)
        cond[:, :tmp] = prompt_mel
        x = (self.solve if self.use_engine else self.solve_ops)(z.clone(), mu, spk_e, cond)
        return x[:, tmp:].contiguous()

    def decode_ragged(self, tokens: List[torch.Tensor], prompt_mels: List[torch.Tensor], spk: torch.Tensor,
                      zs: List[torch.Tensor]) -> List[torch.Tensor]:
        """Ragged batch: row i = (prompt + generated tokens [Ti], prompt mel [Tm_p_i, mel], noise z [mel_total_i, mel]).
        Padded to the longest row, every stage masked by the row lengths -> list of mels [mel_total_i - Tm_p_i, mel],
        each equal to running that utterance alone."""
        cfg, dev = self.cfg, self.device
        b = len(tokens)
        tl = [int(t.numel()) for t in tokens]
        tmp = [int(m.shape[0]) for m in prompt_mels]
        mt = [int(z.shape[0]) for z in zs]
        tmax, mmax = max(tl), max(mt)
        tok = torch.zeros((b, tmax), dtype=torch.int32, device=dev)
        cond = torch.zeros((b, mmax, cfg.mel), dtype=torch.float32, device=dev)
        x = torch.zeros((b, mmax, cfg.mel), dtype=torch.float32, device=dev)
        for i in range(b):
            tok[i, :tl[i]] = tokens[i].to(dev).view(-1)
            cond[i, :tmp[i]] = prompt_mels[i].to(dev)
            x[i, :mt[i]] = zs[i].to(dev)
        tok_lens = torch.tensor(tl, dtype=torch.int32, device=dev)
        mel_lens = torch.tensor(mt, dtype=torch.int32, device=dev)
        mu = self.mu(tok, tok_lens, mmax, mel_lens)
        spk_e = ops.linear(torch.nn.functional.normalize(spk.to(dev), dim=1), self.spk_aff)
        x = (self.solve if self.use_engine else self.solve_ops)(x, mu, spk_e, cond, mel_lens)
        return [x[i, tmp[i]:mt[i]].contiguous() for i in range(b)]


class _ResBlock:
    def __init__(self, sd: SD, p: str, k: int, dils, device):
        self.k, self.dils = k, dils
        self.c1 = [PackedWeight.from_conv1d(sd[f"{p}.convs1.{j}.weight"], sd[f"{p}.convs1.{j}.bias"], device) for j in range(len(dils))]
        self.c2 = [PackedWeight.from_conv1d(sd[f"{p}.convs2.{j}.weight"], sd[f"{p}.convs2.{j}.bias"], device) for j in range(len(dils))]
        self.a1 = [_dev(sd[f"{p}.activations1.{j}.alpha"], device) for j in range(len(dils))]
        self.a2 = [_dev(sd[f"{p}.activations2.{j}.alpha"], device) for j in range(len(dils))]
        # LDS-staged form (ops.conv1d_snake: Snake applied while the input tile is staged, taps as shifted LDS reads, bias /
        # residual / resblock mean in the epilogue) for the production widths (128 / 256 channels)
        c = self.c1[0].n
        self.lds = all(w.n == c and w.cin == c and w.cin == w.cin_pad for w in self.c1 + self.c2) and \
            all(ops.conv1d_snake_supported(c, k, d) for d in dils)
        self.f1 = [ops.conv_pack_frag(w) for w in self.c1] if self.lds else None
        self.f2 = [ops.conv_pack_frag(w) for w in self.c2] if self.lds else None

    def forward(self, x: torch.Tensor, acc: Optional[torch.Tensor] = None, acc_scale: float = 1.0, acc_add: bool = False,
                lens: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """x + the branches, iteration by iteration: x <- x + conv2(snake(conv1(snake(x)))).  With ``acc`` the last iteration writes
        ``acc = (acc if acc_add else 0) + acc_scale * x_final`` instead of returning ``x_final`` (the mean over the parallel
        resblocks / the sum with the up-sampled stream without a launch of their own)."""
        k = self.k
        n = len(self.dils)
        for j, d in enumerate(self.dils):
            last = j == n - 1
            if self.lds:
                h = ops.conv1d_snake(x.contiguous(), self.c1[j], self.f1[j], dil=d, alpha=self.a1[j], out_dtype=torch.float16, lens=lens)
                if last and acc is not None:
                    ops.conv1d_snake(h, self.c2[j], self.f2[j], dil=1, alpha=self.a2[j], residual=x, want_y=False, acc=acc,
                                     acc_scale=acc_scale, acc_add=acc_add, lens=lens)
                    return None
                x = ops.conv1d_snake(h, self.c2[j], self.f2[j], dil=1, alpha=self.a2[j], residual=x, lens=lens)
            else:
                xt = ops.elementwise(ops.EL_SNAKE, x, p0=self.a1[j])
                xt = ops.conv1d(xt, self.c1[j], dil=d, pad=d * (k - 1) // 2, lens=lens)
                xt = ops.elementwise(ops.EL_SNAKE, xt, p0=self.a2[j])
                x = ops.conv1d(xt, self.c2[j], pad=(k - 1) // 2, residual=x, lens=lens)
        if acc is not None:
            if acc_add:
                acc.copy_(ops.elementwise(ops.EL_ADD, acc, z=x, s=acc_scale))
            else:
                acc.copy_(ops.elementwise(ops.EL_SCALE, x, s=acc_scale))
            return None
        return x


class HiftVocoder:
    """HiFTGenerator: f0 predictor -> NSF source -> STFT -> conv-transpose / Snake-resblock stack -> iSTFT."""

    def __init__(self, sd: SD, cfg: SynthConfig, device):
        self.cfg, self.device = cfg, device
        self.f0_convs = [PackedWeight.from_conv1d(sd[f"f0_predictor.condnet.{2 * j}.weight"], sd[f"f0_predictor.condnet.{2 * j}.bias"], device) for j in range(5)]
        self.f0_cls = PackedWeight(sd["f0_predictor.classifier.weight"], sd["f0_predictor.classifier.bias"], device)
        self.src_w = _dev(sd["m_source.l_linear.weight"].reshape(-1), device)
        self.src_b = _dev(sd["m_source.l_linear.bias"].reshape(-1), device)
        self.conv_pre = PackedWeight.from_conv1d(sd["conv_pre.weight"], sd["conv_pre.bias"], device)
        self.ups, self.sdowns, self.sres, self.res = [], [], [], []
        nk = len(cfg.res_kernels)
        for i, r in enumerate(cfg.up_rates):
            self.ups.append(PackedWeight.from_conv_transpose1d(sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], r, device))
            wd = sd[f"source_downs.{i}.weight"]
            self.sdowns.append((PackedWeight.from_conv1d(wd, sd[f"source_downs.{i}.bias"], device), int(wd.shape[-1])))
            self.sres.append(_ResBlock(sd, f"source_resblocks.{i}", cfg.src_res_kernels[i], cfg.res_dils, device))
            self.res.append([_ResBlock(sd, f"resblocks.{i * nk + kk}", k, cfg.res_dils, device) for kk, k in enumerate(cfg.res_kernels)])
        self.conv_post = PackedWeight.from_conv1d(sd["conv_post.weight"], sd["conv_post.bias"], device)

    def f0(self, mel: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = mel
        for w in self.f0_convs:
            h = ops.conv1d(h, w, pad=1, act="elu", lens=lens)
        f = ops.linear(h, self.f0_cls)
        return torch.abs(f.squeeze(-1))  # |.| of a [B, Tm] vector: plumbing

    def source(self, f0: torch.Tensor, phase0: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        return ops.nsf_source(f0.contiguous(), phase0, noise, self.src_w, self.src_b, cfg.upsample_total,
                              float(cfg.sample_rate), cfg.nsf_alpha, cfg.nsf_sigma, cfg.nsf_voiced_threshold)

    def decode(self, mel: torch.Tensor, source: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``lens`` (int32 ``[B]``, mel frames of each row): a RAGGED batch in one pass.  Every convolution reads a row as a sequence of
        its own length (zero padding behind it), the STFT mirrors a row's own last samples and the iSTFT overlap-adds its own frames
        only, so the first ``lens[b] * upsample_total`` samples of row b equal the row run alone (the reference runs every utterance
        alone: tts_with_rag.py:172-197); what lies behind them is unspecified."""
        cfg = self.cfg
        up = cfg.upsample_total
        ln = (lambda scale, add=0: None) if lens is None else (lambda scale, add=0: (lens * scale + add).to(torch.int32))
        s_stft = ops.stft16(source, lens=ln(up))                             # [B, F, 18]
        x = ops.conv1d(mel, self.conv_pre, pad=3, lens=ln(1))
        n_up = len(cfg.up_rates)
        nk = len(cfg.res_kernels)
        rate = 1
        for i, r in enumerate(cfg.up_rates):
            x = ops.elementwise(ops.EL_LEAKY, x, s=cfg.lrelu_slope)
            x = ops.conv_transpose1d(x, self.ups[i], padding=r // 2, lens=ln(rate))
            rate *= r
            extra = 0
            if i == n_up - 1:
                x = torch.cat([x[:, 1:2], x], dim=1)                           # ReflectionPad1d((1, 0)): plumbing
                extra = 1
            wd, kd = self.sdowns[i]
            f_lens = ln(up // 4, 1)                                            # STFT frames of each row
            si = ops.conv1d(s_stft, wd, stride=kd // 2, pad=kd // 4, lens=f_lens) if kd > 1 else ops.conv1d(s_stft, wd, lens=f_lens)
            x = x.contiguous()
            cur = ln(rate, extra)
            self.sres[i].forward(si, acc=x, acc_scale=1.0, acc_add=True, lens=cur)      # x += source resblock(si): the last conv's epilogue
            xs = torch.empty_like(x) if lens is None else torch.zeros_like(x)   # (tiles behind a short row's end are skipped: keep them finite)
            for kk, rb in enumerate(self.res[i]):                            # mean of the parallel resblocks, same way
                rb.forward(x, acc=xs, acc_scale=1.0 / nk, acc_add=kk > 0, lens=cur)
            x = xs
        x = ops.elementwise(ops.EL_LEAKY, x, s=0.01)
        x = ops.conv1d(x, self.conv_post, pad=3, lens=ln(up // 4, 1))
        return ops.istft16(x, 100.0, cfg.audio_limit, frame_lens=ln(up // 4, 1))

    def forward(self, mel, phase0, noise, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.decode(mel, self.source(self.f0(mel, lens), phase0, noise), lens)

    def forward_ragged(self, mels: List[torch.Tensor], phase0s: List[torch.Tensor], noises: List[torch.Tensor]) -> List[torch.Tensor]:
        """Utterances of different lengths in ONE vocoder pass (``mels[j]``: ``[Tm_j, 80]``, ``phase0s[j]``: ``[1, H+1]``, ``noises[j]``:
        ``[1, Tm_j * upsample_total, H+1]``) -> waveforms ``[1, Tm_j * upsample_total]``, each equal to ``forward`` of the row alone."""
        cfg, dev = self.cfg, self.device
        b, up = len(mels), cfg.upsample_total
        tl = [int(m.shape[0]) for m in mels]
        tmax = max(tl)
        mel = torch.zeros((b, tmax, cfg.mel), dtype=torch.float32, device=dev)
        noise = torch.zeros((b, tmax * up, cfg.nb_harmonics + 1), dtype=torch.float32, device=dev)
        for j in range(b):
            mel[j, :tl[j]] = mels[j]
            noise[j, :tl[j] * up] = noises[j].reshape(tl[j] * up, -1)
        lens = torch.tensor(tl, dtype=torch.int32, device=dev)
        wav = self.forward(mel, torch.cat([p.reshape(1, -1) for p in phase0s], 0).to(dev), noise, lens)
        return [wav[j:j + 1, :tl[j] * up] for j in range(b)]


class SynthEngine:
    """All three stages on one GPU."""

    def __init__(self, state: Dict[str, SD], cfg: SynthConfig, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("astts.synth needs a ROCm GPU; there is no CPU fallback in the product path")
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):
            self.lm = AcousticLM(state["llm"], cfg, self.device)
            self.flow = FlowDecoder(state["flow"], cfg, self.device)
            self.hift = HiftVocoder(state["hift"], cfg, self.device)

    def tts(self, text, text_lens, lm_spk, lm_prompt_tokens, n_tokens: int, uniforms, flow_prompt_tokens, flow_prompt_mel,
            flow_spk, z, phase0, noise, forced_tokens=None, wide: bool = False):
        """One fixed-length batch end to end (all inputs on the GPU):
        LM decode (style-conditioned) -> flow (timbre-conditioned) -> vocoder.  Returns
        (tokens [B, n_tokens], mel [B, Tm, 80], wav [B, 256*Tm])."""
        toks = self.tts_tokens(text, text_lens, lm_spk, lm_prompt_tokens, n_tokens, uniforms, forced_tokens, wide)
        mel, wav = self.tts_render(toks, flow_prompt_tokens, flow_prompt_mel, flow_spk, z, phase0, noise)
        return toks, mel, wav

    # the two halves of tts(): the latency-bound autoregressive stage and the throughput-bound rendering stage
    def tts_tokens(self, text, text_lens, lm_spk, lm_prompt_tokens, n_tokens: int, uniforms, forced_tokens=None, wide: bool = False) -> torch.Tensor:
        pre = self.lm.prefix(text, text_lens, lm_spk, lm_prompt_tokens)
        return self.lm.decode(pre, n_tokens, uniforms, ignore_eos=True, forced_tokens=forced_tokens, wide=wide)

    # the same in two halves (<= 32 rows): what precedes the first sampled token, and the decode steps (PipelinedSynth)
    def tts_prefill(self, text, text_lens, lm_spk, lm_prompt_tokens, n_tokens: int):
        return self.lm.prefill(self.lm.prefix(text, text_lens, lm_spk, lm_prompt_tokens), n_tokens)

    def tts_decode(self, state, uniforms, forced_tokens=None) -> torch.Tensor:
        return self.lm.decode_prefilled(state, uniforms, ignore_eos=True, forced_tokens=forced_tokens)

    def tts_render(self, toks, flow_prompt_tokens, flow_prompt_mel, flow_spk, z, phase0, noise):
        cfg = self.cfg
        all_tok = torch.cat([flow_prompt_tokens.to(torch.int32), toks], dim=1)
        b = all_tok.shape[0]
        tok_lens = torch.full((b,), all_tok.shape[1], dtype=torch.int32, device=self.device)
        mel_total = flow_prompt_mel.shape[1] + cfg.mel_frames_for_tokens(toks.shape[1])
        mel = self.flow.decode(all_tok, tok_lens, flow_prompt_mel, flow_spk, z, mel_total)
        wav = self.hift.forward(mel, phase0, noise)
        return mel, wav


def _lib_check_spin(us: int, stream) -> None:
    from .. import _lib
    _lib.check(_lib.load().astts_stream_spin(int(us), int(stream.cuda_stream)))


class PipelinedSynth:
    """Software pipeline over consecutive (independent) batches on several HIP streams of one GPU.

    The LM decode is a chain of ~19k small dependent launches (latency-bound: it leaves most CUs idle), the
    flow + vocoder stage is throughput-bound.  ``lm_depth`` decode chains run concurrently on their own high-priority
    streams, each enqueued by its own host thread (the decode loop is C++: ctypes drops the GIL), while the render
    stage of an earlier batch runs on a further stream; an event hands each batch's tokens over.
    ``submit`` enqueues batch i and returns the (toks, mel, wav) of the batch whose render stage it enqueued
    (None while the pipeline fills); ``drain`` enqueues what is left and returns those results.  Nothing here
    synchronises the host with the GPU.

    ``cobatch`` > 1: the LM stages of that many consecutive compatible batches (same prefix / decode lengths, <= 32 rows
    together) run as ONE decode chain with their rows side by side -- the chain's ~19k launches are latency-bound and cost
    about the same for 16 rows as for 8 -- and each batch is then rendered on its own.  Rows are independent in every LM
    kernel (GEMM rows, per-(row, head) attention, per-row sampler with per-row uniforms), so every batch's tokens, mel and
    waveform are bit-identical to running it alone (tested)."""

    def __init__(self, engine: "SynthEngine", lm_depth: int = 2, lm_priority: int = -1, render_priority: int = 0, streams=None,
                 cobatch: int = 1, render_depth: int = 1, pipe_classes=None, front_prefill: Optional[bool] = None,
                 stagger_ms: Optional[float] = None, wide_lm: bool = False):
        import os
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        self.eng = engine
        self.depth = max(1, int(lm_depth))
        # ``render_depth`` > 1: the flow + vocoder stages of consecutive batches alternate between that many streams.  One
        # render chain is ~6000 dependent launches of ~10 us that each fill the chip for a few microseconds and leave it
        # to latency (tails, boundaries) in between; a second chain runs in those gaps.
        self.render_depth = rd = max(1, int(render_depth))
        dev = engine.device
        with torch.cuda.device(dev):
            self.front_stream = None
            extra_render = []
            if streams is not None:
                self.s_lm, self.s_render = list(streams[:self.depth]), streams[self.depth]
                extra_render = list(streams[self.depth + 1:self.depth + rd])
            elif lm_priority == render_priority and rd > 1:
                st = ops.concurrent_streams(self.depth + 1 + rd, priority=lm_priority, device=dev)
                self.s_render, extra_render, self.front_stream, self.s_lm = st[0], st[1:rd], st[rd], st[rd + 1:]
            elif lm_priority == render_priority:
                # One stream per command-processor pipe (ops.stream_pipe_classes: launch chains whose queues share a pipe take
                # turns at every kernel boundary and run 2.4x slower each), in order of importance: the render stream, the
                # decode chains, then a front stream for the caller's own per-batch work (retrieval, input preparation: run
                # `submit` under `torch.cuda.stream(pipe.front_stream)`).  With more streams than pipes the front stream (a
                # few launches per batch) doubles up with the render stream's pipe, then decode chains double up among
                # themselves -- never with the render stream.
                classes = pipe_classes if pipe_classes is not None else ops.stream_pipe_classes(priority=lm_priority, device=dev)
                firsts = [c[0] for c in classes]
                need = self.depth + 1
                if len(firsts) >= need + 1:
                    self.s_render, self.s_lm, self.front_stream = firsts[0], firsts[1:need], firsts[need]
                else:
                    self.s_render = firsts[0]
                    self.front_stream = classes[0][1] if len(classes[0]) > 1 else None
                    lm_classes = classes[1:] if len(classes) > 1 else classes
                    self.s_lm = []
                    for i in range(self.depth):
                        cl = lm_classes[i % len(lm_classes)]
                        j = i // len(lm_classes) + (0 if len(classes) > 1 else 2)
                        self.s_lm.append(cl[j] if j < len(cl) else torch.cuda.Stream(device=dev, priority=lm_priority))
            else:
                self.s_lm = [torch.cuda.Stream(device=dev, priority=lm_priority) for _ in range(self.depth)]
                self.s_render = torch.cuda.Stream(device=dev, priority=render_priority)
            if self.front_stream is None:
                self.front_stream = torch.cuda.Stream(device=dev)
            while len(extra_render) < rd - 1:
                extra_render.append(torch.cuda.Stream(device=dev, priority=render_priority))
            self.s_renders = [self.s_render] + list(extra_render)
            self._r = 0
        self._pool = ThreadPoolExecutor(max_workers=self.depth)
        self._fifo = deque()
        self._pending = []                  # batches waiting for their (co-batched) LM stage to be launched
        self.cobatch = max(1, int(cobatch))
        # ``wide_lm`` (throughput runs): LM stages of more than 32 rows decode as ONE chain on the engine's wide path (plain GEMMs, the
        # weights read once per token for all rows) instead of 32-row groups; see AcousticLM.decode
        self.wide_lm = bool(wide_lm)
        self.lm_rows_max = AcousticLM.WIDE_ROWS if self.wide_lm else 32
        self.front_prefill = (os.environ.get("ASTTS_PIPE_FRONT_PREFILL", "1") != "0") if front_prefill is None else bool(front_prefill)
        self.stagger_ms = float(os.environ.get("ASTTS_PIPE_STAGGER_MS", "0")) if stagger_ms is None else float(stagger_ms)
        self._staggered = set()
        self._i = 0

    @classmethod
    def autotune(cls, engine: "SynthEngine", sample_args, depths=(3, 2), trials: int = 3, steps: int = 4, verbose: bool = False,
                 front=None, dist=None, wide_lm: bool = False):
        """Build the pipeline by measurement.  How well the chains overlap depends on which hardware queues HIP hands the
        streams (it multiplexes streams onto a few queues in an order the caller cannot see; a chain that shares a queue
        with another busy stream, or with the stream the caller enqueues its own work on, stalls the hand-over events).
        Each trial draws a fresh set of mutually concurrent streams (ops.concurrent_streams), runs ``steps`` batches of
        ``sample_args`` (the positional arguments of ``submit``) and the fastest configuration is kept.  ``front``: the
        caller's own per-batch GPU work (a callable, e.g. the style retrieval), enqueued on ``pipe.front_stream`` before
        each submit exactly as the caller will.  ``dist`` (an initialised torch.distributed module): every rank runs the same
        trials and ALL ranks keep the configuration whose slowest rank is fastest (one all-reduce MAX of the per-configuration
        times) -- ranks that picked different chain counts would straggle at the per-step all-gather of the style ids."""
        import os
        import time

        best, best_dt = None, float("inf")
        per_cfg = []                        # (best pipe, best time) per configuration, in the order of `depths`
        with torch.cuda.device(engine.device):
            classes = ops.stream_pipe_classes(device=engine.device, verbose=verbose)      # one probe for all trials
        for cfg_ in depths:
            cfg_ = cfg_ if isinstance(cfg_, tuple) else (cfg_, 1)
            depth, cob, rdep = (tuple(cfg_) + (1,))[:3]
            for _ in range(trials):
                lm_prio = int(os.environ.get("ASTTS_PIPE_LM_PRIORITY", "0"))      # experiments: -1 = high-priority decode streams
                pipe = cls(engine, lm_depth=depth, lm_priority=lm_prio, render_priority=0, cobatch=cob, render_depth=rdep,
                           pipe_classes=classes if rdep == 1 and lm_prio == 0 else None, wide_lm=wide_lm)
                with torch.cuda.stream(pipe.front_stream):
                    for _ in range((depth + 1) * cob):
                        if front is not None:
                            front()
                        pipe.submit(*sample_args)
                    pipe.drain()
                    torch.cuda.synchronize(engine.device)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        if front is not None:
                            front()
                        pipe.submit(*sample_args)
                    pipe.drain()
                    torch.cuda.synchronize(engine.device)
                dt = (time.perf_counter() - t0) / steps
                if verbose:
                    print(f"PipelinedSynth.autotune: depth {depth} cobatch {cob} render streams {rdep}: {dt * 1e3:.1f} ms/batch", flush=True)
                if dt < best_dt:
                    best, best_dt = pipe, dt
                if not per_cfg or per_cfg[-1][2] != cfg_:
                    per_cfg.append([pipe, dt, cfg_])
                elif dt < per_cfg[-1][1]:
                    per_cfg[-1][0], per_cfg[-1][1] = pipe, dt
        if dist is not None and len(per_cfg) > 1:
            t = torch.tensor([c[1] for c in per_cfg], dtype=torch.float64, device=engine.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            k = int(torch.argmin(t).item())
            best, best_dt = per_cfg[k][0], per_cfg[k][1]
            best.tuned_agreed_over_ranks = True
        best.tuned_ms_per_batch = best_dt * 1e3
        best.tuned_table_ms = {str(c[2]): round(c[1] * 1e3, 2) for c in per_cfg}
        return best

    # ---- stages
    def _launch_group(self):
        """Start the LM stage of the pending batches as ONE decode chain (rows of all of them side by side)."""
        group, self._pending = self._pending, []
        chain = self._i % self.depth
        stream = self.s_lm[chain]
        self._i += 1
        cur = torch.cuda.current_stream(self.eng.device)
        # ``stagger_ms``: after an idle pipeline (construction, drain) chain c starts its first decode c * stagger_ms late (a one-wave
        # busy-wait kernel at the head of its stream).  Chains that start together stay in phase: their batches finish together, the
        # render stage gets them in bursts and the drain ends with `depth` renders in a row; offset by a third of a chain's period they
        # hand the render stage one batch at a time and the drain ends with ONE render.
        if self.stagger_ms > 0 and chain > 0 and chain not in self._staggered:
            self._staggered.add(chain)
            us = int(self.stagger_ms * 1000.0 * chain)
            while us > 0:
                _lib_check_spin(min(us, 100000), stream)
                us -= 100000
        if len(group) == 1:
            lm_args = group[0]["lm"]
        else:   # rows are independent in every LM kernel: a row's tokens do not depend on which rows sit next to it
            a = [g["lm"] for g in group]
            lm_args = (torch.cat([x[0] for x in a], 0), torch.cat([x[1] for x in a], 0), torch.cat([x[2] for x in a], 0),
                       torch.cat([x[3] for x in a], 0), a[0][4], torch.cat([x[5] for x in a], 1))
        # every input -- the caller's own tensors too -- was allocated on the caller's stream and is read on the chain's for
        # the whole decode (~100 ms behind the host): tell the caching allocator, or a caller that builds fresh inputs per
        # batch gets these blocks recycled under the running chain
        for t in lm_args:
            if isinstance(t, torch.Tensor):
                t.record_stream(stream)
        sizes = [int(g["lm"][0].shape[0]) for g in group]
        # ``front_prefill`` (default; ASTTS_PIPE_FRONT_PREFILL=0 or front_prefill=False: the whole LM stage on the chain): prefix
        # assembly + prefill (text encoder, 14 layers over the prefix, first logits: ~4.7 ms of a chain's ~106 ms per batch, big-tile
        # GEMM launches) are enqueued HERE, on the caller's stream, and the chain is its decode steps only.  The chains are what
        # bounds the pipelined step: 95.5 -> 93.9 ms per batch (scripts/front_prefill_probe.py, alternating in one process, 4 of 4)
        state = None
        if self.front_prefill and sum(sizes) <= self.lm_rows_max:
            state = self.eng.tts_prefill(*lm_args[:5])
            for t in self.eng.lm.prefill_tensors(state):
                t.record_stream(stream)
        stream.wait_stream(cur)             # after the concatenations (and the prefill) above were enqueued

        def lm_stage():
            with torch.cuda.device(self.eng.device), torch.cuda.stream(stream):
                toks = self.eng.tts_tokens(*lm_args, wide=self.wide_lm) if state is None else self.eng.tts_decode(state, lm_args[5])
                parts = list(torch.split(toks, sizes, 0)) if len(sizes) > 1 else [toks]
                ev = torch.cuda.Event()
                ev.record(stream)
            return parts, ev

        fut = self._pool.submit(lm_stage)
        for k, g in enumerate(group):
            g["fut"], g["k"] = fut, k

    def _render(self, item):
        parts, ev = item["fut"].result()
        toks = parts[item["k"]]
        cur = torch.cuda.current_stream(self.eng.device)
        sr = self.s_renders[self._r % self.render_depth]
        self._r += 1
        with torch.cuda.stream(sr):
            sr.wait_event(ev)
            toks.record_stream(sr)
            for t in item["render"]:        # the caller's tensors, read on the render stream (see _launch_group)
                if isinstance(t, torch.Tensor):
                    t.record_stream(sr)
            mel, wav = self.eng.tts_render(toks, *item["render"])
        # results are produced on the pipeline's streams and handed to the caller's: the caller still has to order its
        # stream behind them (drain() does; a caller that consumes results earlier waits on `pipe.s_render` itself), but
        # the allocator must not recycle them while the caller's stream may be using them
        for t in (toks, mel, wav):
            t.record_stream(cur)
        return toks, mel, wav

    def submit(self, text, text_lens, lm_spk, lm_prompt_tokens, n_tokens, uniforms, flow_prompt_tokens, flow_prompt_mel,
               flow_spk, z, phase0, noise):
        cur = torch.cuda.current_stream(self.eng.device)
        for sr in self.s_renders:
            sr.wait_stream(cur)
        item = {"lm": (text, text_lens, lm_spk, lm_prompt_tokens, n_tokens, uniforms),
                "render": (flow_prompt_tokens, flow_prompt_mel, flow_spk, z, phase0, noise), "fut": None, "k": 0}
        if self._pending and not self._compatible(self._pending[0]["lm"], item["lm"]):
            self._launch_group()
        self._pending.append(item)
        self._fifo.append(item)
        if len(self._pending) >= self.cobatch:
            self._launch_group()
        # keep `depth` decode chains in flight behind the batch being rendered
        if len(self._fifo) > self.depth * self.cobatch and self._fifo[0]["fut"] is not None:
            return self._render(self._fifo.popleft())
        return None

    def _compatible(self, a, b) -> bool:
        """Batches can share a decode chain when their prefix and decode lengths agree (fixed-length batches of one job)."""
        return (a[4] == b[4] and a[0].shape[1] == b[0].shape[1] and a[3].shape[1] == b[3].shape[1] and a[5].shape[0] == b[5].shape[0]
                and a[0].shape[0] + b[0].shape[0] <= self.lm_rows_max)

    def drain(self):
        if self._pending:
            self._launch_group()
        self._staggered = set()          # the pipeline runs empty: the next first batches are staggered again
        self._i = 0
        out = []
        while self._fifo:
            out.append(self._render(self._fifo.popleft()))
        cur = torch.cuda.current_stream(self.eng.device)
        for st in self.s_lm:
            cur.wait_stream(st)
        for sr in self.s_renders:
            cur.wait_stream(sr)
        return out
