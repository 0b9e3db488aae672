"""CPU oracle for the synthesis path: fp32 PyTorch restatement.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py for who may import this).

PARITY UNPINNED.  The arithmetic behind the reference's ``cosyvoice.inference_tts_with_st`` /
``inference_zero_shot`` / ``inference_vc`` calls (/root/reference/tts_with_rag.py:195,133,141;
tts_with_style_and_timbre.py:93,47,57) lives in the authors' private CosyVoice fork + Matcha-TTS,
neither vendored nor importable here, and no weights exist in this environment.  This file restates
the *published* CosyVoice-300M architecture ([EXT]-recalled):
    TransformerLM     espnet rel-pos text encoder -> causal rel-pos transformer with KV cache -> RAS sampling
    MaskedDiffWithXvec token encoder -> length regulator -> conditional flow matching (Euler, CFG) with a
                      1-D U-Net estimator (ResnetBlock1D + BasicTransformerBlock)
    HiFTGenerator     f0 predictor -> NSF harmonic source -> conv-transpose / Snake-resblock stack -> iSTFT
and is the checker for the HIP path under identical weights (astts.synth.weights) and identical
INJECTED randomness (sampling uniforms, CFM z, source phases/noise).  Tensors are channels-last.
"""
from __future__ import annotations

import functools
import math
import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------- shared
def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd: SD, p: str, x: torch.Tensor, stride=1, dil=1, pad=0) -> torch.Tensor:
    """x [B, T, C] -> [B, T', C']"""
    return F.conv1d(x.transpose(1, 2), sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, dilation=dil,
                    padding=pad).transpose(1, 2)


def _mask(lens: torch.Tensor, t: int) -> torch.Tensor:
    return (torch.arange(t)[None, :] < lens[:, None]).float()[..., None]  # [B, T, 1]


@functools.lru_cache(maxsize=8)
def rel_pos_table(d: int, max_pos: int) -> torch.Tensor:
    """espnet relative positional encoding rows for rel = -max_pos..max_pos: pe[2i] = sin(rel*w_i),
    pe[2i+1] = cos(rel*w_i).  Row index = rel + max_pos.  (Memoised: a constant of (d, max_pos); callers only read it.)"""
    rel = torch.arange(-max_pos, max_pos + 1, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * max_pos + 1, d)
    pe[:, 0::2] = torch.sin(rel * div)
    pe[:, 1::2] = torch.cos(rel * div)
    return pe


_PTAB: Dict[int, tuple] = {}


def _linear_pos(w: torch.Tensor, pe: torch.Tensor) -> torch.Tensor:
    """linear_pos(pe): the layer's projected position table.  It depends on the layer's weight and the table only, so it is formed
    once per (weight tensor, table) and reused -- identical arithmetic, but a decode step no longer re-projects all 2c+1 rows in
    every layer (8.6 GFLOP per call at c = 2048, d = 1024: what upstream avoids by projecting only the rows a call needs)."""
    ent = _PTAB.get(id(w))
    if ent is not None and ent[0]() is w and ent[1] is pe and ent[2] == w._version:
        return ent[3]
    out = F.linear(pe, w)
    if len(_PTAB) > 256:
        _PTAB.clear()
    _PTAB[id(w)] = (weakref.ref(w), pe, w._version, out)
    return out


def relpos_attention(sd: SD, p: str, x: torch.Tensor, heads: int, pe: torch.Tensor, center: int,
                     lens: torch.Tensor, causal: bool, cache: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """RelPositionMultiHeadedAttention: score(i,j) = ((q_i+u).k_j + (q_i+v).P(i-j)) / sqrt(dk).
    ``x`` are the NEW positions; with ``cache`` (k, v of earlier positions) keys = cache + new."""
    b, tq, d = x.shape
    dk = d // heads
    q = _lin(sd, p + ".linear_q", x)
    k = _lin(sd, p + ".linear_k", x)
    v = _lin(sd, p + ".linear_v", x)
    if cache is not None:
        k = torch.cat([cache[0], k], dim=1)
        v = torch.cat([cache[1], v], dim=1)
    tk = k.shape[1]
    q_pos0 = tk - tq
    ptab = _linear_pos(sd[p + ".linear_pos.weight"], pe)          # [2c+1, d]
    qh = q.view(b, tq, heads, dk).transpose(1, 2)
    kh = k.view(b, tk, heads, dk).transpose(1, 2)
    vh = v.view(b, tk, heads, dk).transpose(1, 2)
    i = torch.arange(tq)[:, None] + q_pos0
    j = torch.arange(tk)[None, :]
    pr = ptab[(i - j) + center].view(tq, tk, heads, dk).permute(2, 0, 1, 3)
    ac = torch.einsum("bhid,bhjd->bhij", qh + sd[p + ".pos_bias_u"].view(1, heads, 1, dk), kh)
    bd = torch.einsum("bhid,hijd->bhij", qh + sd[p + ".pos_bias_v"].view(1, heads, 1, dk), pr)
    s = (ac + bd) / math.sqrt(dk)
    m = j[None] < lens[:, None, None]
    if causal:
        m = m & (j <= i)[None]
    s = s.masked_fill(~m[:, None], float("-inf"))
    o = torch.einsum("bhij,bhjd->bhid", torch.softmax(s, dim=-1), vh).transpose(1, 2).reshape(b, tq, d)
    return _lin(sd, p + ".linear_out", o), (k, v)


def relpos_encoder(sd: SD, p: str, x: torch.Tensor, lens: torch.Tensor, heads: int, layers: int, act: str,
                   norm_names: Tuple[str, str], legacy_embed: bool, causal: bool, eps: float, max_pos: int,
                   caches: Optional[List] = None):
    """embed (Linear+LayerNorm[+ReLU]) * sqrt(d) -> pre-norm layers -> after_norm.
    Returns (y, new_caches)."""
    d = sd[p + ".after_norm.weight"].shape[0]
    h = _ln(sd, p + ".embed.out.1", _lin(sd, p + ".embed.out.0", x), eps)
    if legacy_embed:
        h = F.relu(h)
    h = h * math.sqrt(d)
    pe = rel_pos_table(d, max_pos)
    fn = {"relu": F.relu, "swish": F.silu}[act]
    new_caches = []
    for i in range(layers):
        h, kv = relpos_layer(sd, f"{p}.encoders.{i}", h, heads, pe, max_pos, lens, causal, norm_names, fn, eps,
                             None if caches is None else caches[i])
        new_caches.append(kv)
    return _ln(sd, p + ".after_norm", h, eps), new_caches


def relpos_layer(sd: SD, q: str, h: torch.Tensor, heads: int, pe: torch.Tensor, center: int, lens: torch.Tensor, causal: bool,
                 norm_names: Tuple[str, str], fn, eps: float, cache=None):
    """One pre-norm encoder layer (no macaron feed-forward, no convolution module): h += attn(LN(h)); h += W2 act(W1 LN(h)).
    Pinned against transformers' FastSpeech2ConformerEncoderLayer (tests/test_oracle_synth_blocks.py)."""
    n1, n2 = norm_names
    a, kv = relpos_attention(sd, q + ".self_attn", _ln(sd, f"{q}.{n1}", h, eps), heads, pe, center, lens, causal, cache)
    h = h + a
    f = _lin(sd, q + ".feed_forward.w_2", fn(_lin(sd, q + ".feed_forward.w_1", _ln(sd, f"{q}.{n2}", h, eps))))
    return h + f, kv


# ----------------------------------------------------------------------------------------- LM
def nucleus(p: torch.Tensor, top_k: int, top_p: float):
    """The candidate set of upstream's nucleus_sampling [EXT cosyvoice/utils/common.py]: probabilities in descending order (ties:
    lower id first), a token is added while the mass of those ALREADY added is < top_p and fewer than top_k were added.
    -> (ids in rank order, how many of them form the nucleus, their fp32 mass accumulated in rank order).
    Pinned against transformers' TopPLogitsWarper (tests/test_oracle_synth_blocks.py): the same set, cut at top_k."""
    order = sorted(range(p.numel()), key=lambda i: (-float(p[i]), i))[:top_k]
    cnt = 0
    cum32 = torch.tensor(0.0)
    for idx in order:
        if float(cum32) < top_p and cnt < top_k:
            cum32 = cum32 + p[idx]
            cnt += 1
        else:
            break
    return order, cnt, cum32


def ras_sample(logits: torch.Tensor, history: torch.Tensor, u: torch.Tensor, top_k: int, top_p: float, win: int,
               tau_r: float, eos: int, ignore_eos: bool, eos_policy: str = "mask") -> torch.Tensor:
    """Repetition-aware sampling with injected uniforms u[b] = (u1, u2); definition in csrc/ops_audio.hip.

    ``ignore_eos``: EOS may not be produced at this step (upstream: step < min_len).  ``eos_policy`` "mask": the EOS logit is removed
    before the softmax.  "reject": upstream's sampling_ids [EXT cosyvoice/llm/llm.py] -- one pass = nucleus draw, repetition check,
    full-distribution draw when the drawn token repeats; passes are repeated with fresh randomness while the result is EOS.  With
    injected uniforms that loop is taken in closed form: conditioned on "not EOS", a pass ends on nucleus entry t (neither EOS nor
    repeated) with weight p_t, or in the fallback with weight (sum of the repeated entries' p) (1 - p_eos); u1 picks among those in rank
    order (fallback last), and the fallback draws with u2 from the distribution without EOS.  Same distribution as the loop."""
    out = []
    reject = ignore_eos and eos_policy == "reject"
    for b in range(logits.shape[0]):
        lg = logits[b].clone().float()
        if ignore_eos and not reject:
            lg[eos] = float("-inf")
        e = torch.exp(lg - lg.max())
        p = e * (1.0 / e.sum())
        order, cnt, cum32 = nucleus(p, top_k, top_p)
        hist = history[b].tolist()[-win:] if history.shape[1] > 0 else []

        def repeated(t):
            return sum(1 for h in hist if h == t) >= win * tau_r

        if not reject:
            target = u[b, 0] * cum32
            run = torch.tensor(0.0)
            tok = order[cnt - 1]
            for idx in order[:cnt]:
                run = run + p[idx]
                if bool(run > target):
                    tok = idx
                    break
            fallback = repeated(tok)
            skip = -1
            target2 = u[b, 1]
        else:
            asum, prep = torch.tensor(0.0), torch.tensor(0.0)
            direct = []
            for idx in order[:cnt]:
                if idx == eos:
                    continue
                if repeated(idx):
                    prep = prep + p[idx]
                else:
                    asum = asum + p[idx]
                    direct.append(idx)
            target = u[b, 0] * (asum + prep * (1.0 - p[eos]))
            run = torch.tensor(0.0)
            tok = -1
            for idx in direct:
                run = run + p[idx]
                if bool(run > target):
                    tok = idx
                    break
            fallback = tok < 0
            skip = eos
            target2 = u[b, 1] * (1.0 - p[eos])
        if fallback:
            run = torch.tensor(0.0)
            pick, last = -1, 0
            for i in range(p.numel()):
                if i == skip:
                    continue
                if float(p[i]) > 0:
                    last = i
                run = run + p[i]
                if bool(run > target2):
                    pick = i
                    break
            tok = pick if pick >= 0 else last
        out.append(tok)
    return torch.tensor(out, dtype=torch.int32)


def lm_prefix(sd: SD, cfg, text: torch.Tensor, text_lens: torch.Tensor, spk: torch.Tensor,
              prompt_tokens: torch.Tensor) -> torch.Tensor:
    """[sos, spk, text_encoder(text), task_id, speech_emb(prompt_tokens)] -> [B, S0, lm_dim].
    All rows of a batch share Tt and Tp here (fixed-length batches; ragged batches are bucketed upstream)."""
    te = sd["text_embedding.weight"][text]
    enc, _ = relpos_encoder(sd, "text_encoder", te, text_lens, cfg.lm_heads, cfg.lm_text_layers, "swish",
                            ("norm_mha", "norm_ff"), False, True, cfg.ln_eps, cfg.max_positions)
    enc = _lin(sd, "text_encoder_affine_layer", enc)
    b = text.shape[0]
    spk_e = _lin(sd, "spk_embed_affine_layer", F.normalize(spk, dim=1))[:, None, :]
    sos = sd["llm_embedding.weight"][0].view(1, 1, -1).expand(b, 1, -1)
    task = sd["llm_embedding.weight"][1].view(1, 1, -1).expand(b, 1, -1)
    pe = sd["speech_embedding.weight"][prompt_tokens]
    return torch.cat([sos, spk_e, enc, task, pe], dim=1)


def lm_forward(sd: SD, cfg, x: torch.Tensor, caches: Optional[List]):
    """Causal LM over new positions ``x`` [B, T, lm_dim] with optional KV caches -> (logits [B,T,V+1], caches)."""
    b, t, _ = x.shape
    tk = t + (0 if caches is None else caches[0][0].shape[1])
    lens = torch.full((b,), tk, dtype=torch.int64)
    y, caches = relpos_encoder(sd, "llm", x, lens, cfg.lm_heads, cfg.lm_layers, "relu", ("norm1", "norm2"), True, True,
                               cfg.ln_eps, cfg.max_positions, caches)
    return _lin(sd, "llm_decoder", y), caches


def lm_decode(sd: SD, cfg, prefix: torch.Tensor, n_steps: int, uniforms: torch.Tensor, ignore_eos: bool = True,
              forced_tokens: Optional[torch.Tensor] = None):
    """Fixed-length autoregressive decode.  ``uniforms`` [n_steps, B, 2].  With ``forced_tokens``
    [B, n_steps] the sampled token is replaced by the forced one after sampling (teacher forcing), so
    logits stay comparable step by step.  Returns (tokens [B, n_steps] int32, logits [B, n_steps, V+1])."""
    b = prefix.shape[0]
    logits, caches = lm_forward(sd, cfg, prefix, None)
    cur = logits[:, -1]
    toks = torch.zeros((b, n_steps), dtype=torch.int32)
    all_logits = []
    for s in range(n_steps):
        all_logits.append(cur)
        tok = ras_sample(cur, toks[:, :s], uniforms[s], cfg.top_k, cfg.top_p, cfg.ras_win, cfg.ras_tau,
                         cfg.speech_vocab, ignore_eos, getattr(cfg, "eos_policy", "mask"))
        if forced_tokens is not None:
            tok = forced_tokens[:, s].to(torch.int32)
        toks[:, s] = tok
        if s + 1 < n_steps:
            emb = sd["speech_embedding.weight"][tok.long().clamp(max=cfg.speech_vocab - 1)][:, None, :]
            lg, caches = lm_forward(sd, cfg, emb, caches)
            cur = lg[:, -1]
    return toks, torch.stack(all_logits, dim=1)


# ----------------------------------------------------------------------------------------- flow
def _block1d(sd: SD, p: str, x: torch.Tensor, m: torch.Tensor, lens: torch.Tensor, groups: int) -> torch.Tensor:
    """matcha Block1D: conv3(x*mask) -> GroupNorm -> Mish, * mask.  GroupNorm statistics are taken over the
    valid frames of each utterance (== the reference run one utterance at a time)."""
    h = _conv(sd, p + ".block.0", x * m, pad=1)
    out = torch.zeros_like(h)
    for i in range(h.shape[0]):
        L = int(lens[i])
        out[i, :L] = F.mish(F.group_norm(h[i:i + 1, :L].transpose(1, 2), groups, sd[p + ".block.1.weight"],
                                         sd[p + ".block.1.bias"], 1e-5)).transpose(1, 2)[0]
    return out


def _resnet1d(sd: SD, p: str, x, m, lens, temb, groups):
    h = _block1d(sd, p + ".block1", x, m, lens, groups)
    h = (h + _lin(sd, p + ".mlp.1", F.mish(temb))[:, None, :]) * m
    h = _block1d(sd, p + ".block2", h, m, lens, groups)
    return h + _conv(sd, p + ".res_conv", x * m)


def _tfm_block(sd: SD, p: str, x, lens, heads, eps=1e-5):
    b, t, c = x.shape
    h = _ln(sd, p + ".norm1", x, eps)
    q, k, v = (F.linear(h, sd[f"{p}.attn1.{n}.weight"]).view(b, t, heads, 64).transpose(1, 2) for n in ("to_q", "to_k", "to_v"))
    mask = (torch.arange(t)[None, :] < lens[:, None])[:, None, None, :]
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(b, t, heads * 64)
    x = x + _lin(sd, p + ".attn1.to_out.0", a)
    h = _ln(sd, p + ".norm3", x, eps)
    return x + _lin(sd, p + ".ff.net.2", F.gelu(_lin(sd, p + ".ff.net.0.proj", h)))


def estimator(sd: SD, cfg, x, mu, spk, cond, t, lens) -> torch.Tensor:
    """ConditionalDecoder: x, mu, cond [B, T, mel]; spk [B, mel]; t [B]; lens [B] -> [B, T, mel]."""
    e = "decoder.estimator"
    b, T, _ = x.shape
    half = cfg.est_in // 2
    emb = 1000.0 * t[:, None] * torch.exp(torch.arange(half).float() * -(math.log(10000.0) / (half - 1)))[None, :]
    temb = torch.cat([emb.sin(), emb.cos()], dim=-1)
    temb = _lin(sd, e + ".time_mlp.linear_2", F.silu(_lin(sd, e + ".time_mlp.linear_1", temb)))
    h = torch.cat([x, mu, spk[:, None, :].expand(b, T, -1), cond], dim=-1)
    hiddens, lens_stack = [], [lens]
    ch = cfg.est_channels
    for i in range(len(ch)):
        p = f"{e}.down_blocks.{i}"
        L = lens_stack[-1]
        m = _mask(L, h.shape[1])
        h = _resnet1d(sd, p + ".0", h, m, L, temb, cfg.est_groups)
        for j in range(cfg.est_tfm_per_block):
            h = _tfm_block(sd, f"{p}.1.{j}", h, L, cfg.est_heads)
        hiddens.append(h)
        if i == len(ch) - 1:
            h = _conv(sd, p + ".2", h * m, pad=1)
            lens_stack.append(L)
        else:
            h = _conv(sd, p + ".2.conv", h * m, stride=2, pad=1)
            lens_stack.append((L + 1) // 2)
    L = lens_stack[-1]
    m = _mask(L, h.shape[1])
    for i in range(cfg.est_mid_blocks):
        p = f"{e}.mid_blocks.{i}"
        h = _resnet1d(sd, p + ".0", h, m, L, temb, cfg.est_groups)
        for j in range(cfg.est_tfm_per_block):
            h = _tfm_block(sd, f"{p}.1.{j}", h, L, cfg.est_heads)
    lens_stack.pop()
    n_up = len(ch)
    for i in range(n_up):
        p = f"{e}.up_blocks.{i}"
        L = lens_stack.pop()
        skip = hiddens.pop()
        m = _mask(L, skip.shape[1])
        h = torch.cat([h[:, :skip.shape[1]], skip], dim=-1)
        h = _resnet1d(sd, p + ".0", h, m, L, temb, cfg.est_groups)
        for j in range(cfg.est_tfm_per_block):
            h = _tfm_block(sd, f"{p}.1.{j}", h, L, cfg.est_heads)
        if i == n_up - 1:
            h = _conv(sd, p + ".2", h * m, pad=1)
        else:
            w = sd[p + ".2.conv.weight"]
            h = F.conv_transpose1d((h * m).transpose(1, 2), w, sd[p + ".2.conv.bias"], stride=2, padding=1).transpose(1, 2)
    m = _mask(lens, T)
    h = _block1d(sd, e + ".final_block", h[:, :T], m, lens, cfg.est_groups)
    return _conv(sd, e + ".final_proj", h * m) * m


def flow_mu(sd: SD, cfg, tokens: torch.Tensor, token_lens: torch.Tensor, mel_total: int):
    """token encoder -> proj -> length regulator: [B, Tp+Ts] tokens -> mu [B, mel_total, mel]."""
    x = sd["input_embedding.weight"][tokens.clamp(min=0)] * _mask(token_lens, tokens.shape[1])
    h, _ = relpos_encoder(sd, "encoder", x, token_lens, cfg.flow_heads, cfg.flow_layers, "swish", ("norm_mha", "norm_ff"),
                          False, False, cfg.ln_eps, cfg.max_positions)
    h = _lin(sd, "encoder_proj", h)
    h = F.interpolate(h.transpose(1, 2), size=mel_total, mode="linear").transpose(1, 2)
    for j in range(4):
        h = _conv(sd, f"length_regulator.model.{3 * j}", h, pad=1)
        h = F.mish(F.group_norm(h.transpose(1, 2), 1, sd[f"length_regulator.model.{3 * j + 1}.weight"],
                                sd[f"length_regulator.model.{3 * j + 1}.bias"], 1e-5)).transpose(1, 2)
    return _conv(sd, "length_regulator.model.12", h)


def cfm_t_grid(n: int) -> torch.Tensor:
    """The n + 1 time points of the Euler solve: the uniform grid warped by the cosine schedule, t -> 1 - cos(pi t / 2).
    Pinned against the sway-sampling grid of transformers' Qwen2_5OmniToken2WavDiTModel.sample at sway_coefficient = -1
    (tests/golden/cfm_grid_cfg.npz, tests/test_oracle_synth_blocks.py)."""
    return 1.0 - torch.cos(torch.linspace(0, 1, n + 1) * 0.5 * math.pi)


def cfg_combine(d_c: torch.Tensor, d_u: torch.Tensor, rate: float) -> torch.Tensor:
    """Classifier-free guidance of the velocity: (1 + rate) conditional - rate unconditional.  Pinned against the same method's
    `guided + (guided - null) * guidance_scale`."""
    return (1.0 + rate) * d_c - rate * d_u


def flow_decode(sd: SD, cfg, tokens, token_lens, prompt_mel, spk, z, mel_total: int) -> torch.Tensor:
    """MaskedDiffWithXvec.inference with injected noise ``z`` [B, mel_total, mel] -> mel [B, mel_total - Tm_p, mel].
    Fixed-length batches: every row uses all mel_total frames."""
    b = tokens.shape[0]
    mu = flow_mu(sd, cfg, tokens, token_lens, mel_total)
    spk_e = _lin(sd, "spk_embed_affine_layer", F.normalize(spk, dim=1))
    tmp = prompt_mel.shape[1]
    cond = torch.zeros(b, mel_total, cfg.mel)
    cond[:, :tmp] = prompt_mel
    lens = torch.full((b,), mel_total, dtype=torch.int64)
    n = cfg.cfm_steps
    ts = cfm_t_grid(n)
    x = z.clone()
    zero = torch.zeros_like(mu)
    for s in range(n):
        t = ts[s].expand(b)
        dt = float(ts[s + 1] - ts[s])
        d_c = estimator(sd, cfg, x, mu, spk_e, cond, t, lens)
        d_u = estimator(sd, cfg, x, zero, torch.zeros_like(spk_e), zero, t, lens)
        x = x + dt * cfg_combine(d_c, d_u, cfg.cfg_rate)
    return x[:, tmp:]


# ----------------------------------------------------------------------------------------- HiFT
def _snake(x, alpha):
    return x + torch.sin(alpha * x) ** 2 / (alpha + 1e-9)


def _resblock(sd: SD, p: str, x, k: int, dils) -> torch.Tensor:
    for j, d in enumerate(dils):
        xt = _snake(x, sd[f"{p}.activations1.{j}.alpha"])
        xt = _conv(sd, f"{p}.convs1.{j}", xt, dil=d, pad=d * (k - 1) // 2)
        xt = _snake(xt, sd[f"{p}.activations2.{j}.alpha"])
        xt = _conv(sd, f"{p}.convs2.{j}", xt, pad=(k - 1) // 2)
        x = xt + x
    return x


def hift_f0(sd: SD, cfg, mel: torch.Tensor) -> torch.Tensor:
    h = mel
    for j in range(5):
        h = F.elu(_conv(sd, f"f0_predictor.condnet.{2 * j}", h, pad=1))
    return torch.abs(_lin(sd, "f0_predictor.classifier", h)).squeeze(-1)  # [B, Tm]


def hift_source(sd: SD, cfg, f0: torch.Tensor, phase0: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """SineGen + SourceModuleHnNSF with injected initial phases [B, H+1] and noise [B, L, H+1] -> [B, L].
    The phase accumulator (cumsum of f0/sr) is evaluated in float64."""
    up = cfg.upsample_total
    nh = cfg.nb_harmonics + 1
    f0u = f0.double().repeat_interleave(up, dim=1)
    harm = torch.arange(1, nh + 1).double()
    theta = 2 * math.pi * ((torch.cumsum(f0u / cfg.sample_rate, dim=1)[..., None] * harm) % 1.0)
    sine = cfg.nsf_alpha * torch.sin(theta.float() + phase0[:, None, :])
    uv = (f0u > cfg.nsf_voiced_threshold).float()[..., None]
    src = sine * uv + (uv * cfg.nsf_sigma + (1 - uv) * cfg.nsf_alpha / 3) * noise
    return torch.tanh(F.linear(src, sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"])).squeeze(-1)


def hift_trunk(sd: SD, cfg, mel: torch.Tensor, s_stft: torch.Tensor) -> torch.Tensor:
    """The HiFi-GAN trunk of HiFT: conv_pre -> per stage [leaky-relu -> ConvTranspose1d -> + source branch -> mean of the
    parallel resblocks] -> leaky-relu(0.01): mel [B, Tm, 80], s_stft [B, F, 18] -> [B, L / 4 + 1, C] (the input of conv_post).
    With a silent source branch and leaky-relu in place of Snake this IS the HiFi-GAN generator trunk: pinned against
    transformers' SpeechT5HifiGan (tests/test_oracle_synth_blocks.py)."""
    x = _conv(sd, "conv_pre", mel, pad=3)
    n_up = len(cfg.up_rates)
    nk = len(cfg.res_kernels)
    for i, r in enumerate(cfg.up_rates):
        x = F.leaky_relu(x, cfg.lrelu_slope)
        x = F.conv_transpose1d(x.transpose(1, 2), sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=r,
                               padding=r // 2).transpose(1, 2)
        if i == n_up - 1:
            x = F.pad(x.transpose(1, 2), (1, 0), mode="reflect").transpose(1, 2)
        wd = sd[f"source_downs.{i}.weight"]
        kd = wd.shape[-1]
        si = _conv(sd, f"source_downs.{i}", s_stft, stride=max(kd // 2, 1), pad=kd // 4) if kd > 1 else _conv(sd, f"source_downs.{i}", s_stft)
        si = _resblock(sd, f"source_resblocks.{i}", si, cfg.src_res_kernels[i], cfg.res_dils)
        x = x + si
        xs = None
        for kk, k in enumerate(cfg.res_kernels):
            y = _resblock(sd, f"resblocks.{i * nk + kk}", x, k, cfg.res_dils)
            xs = y if xs is None else xs + y
        x = xs / nk
    return F.leaky_relu(x)  # default slope 0.01, as upstream


def hift_decode(sd: SD, cfg, mel: torch.Tensor, source: torch.Tensor) -> torch.Tensor:
    """mel [B, Tm, 80], source [B, L] -> waveform [B, L]."""
    win = torch.hann_window(16, periodic=True)
    spec = torch.stft(source, 16, 4, 16, window=win, return_complex=True)
    s_stft = torch.cat([spec.real, spec.imag], dim=1).transpose(1, 2)  # [B, F, 18]
    x = hift_trunk(sd, cfg, mel, s_stft)
    x = _conv(sd, "conv_post", x, pad=3)
    mag = torch.clip(torch.exp(x[..., :9]), max=1e2)
    ph = torch.sin(x[..., 9:])
    cplx = torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)).transpose(1, 2)
    wav = torch.istft(cplx, 16, 4, 16, window=win)
    return wav.clamp(-cfg.audio_limit, cfg.audio_limit)


def hift_forward(sd: SD, cfg, mel, phase0, noise) -> torch.Tensor:
    f0 = hift_f0(sd, cfg, mel)
    return hift_decode(sd, cfg, mel, hift_source(sd, cfg, f0, phase0, noise))
