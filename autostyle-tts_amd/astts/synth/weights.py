"""Seeded synthetic weights at the true CosyVoice-300M shapes (or SynthConfig.tiny()), keyed by the
upstream state-dict names ([EXT]-recalled) so that a real ``llm.pt`` / ``flow.pt`` / ``hift.pt``
loads through the same code path (``load_state_dicts``).  There is no network and no checkpoint in
this environment: benchmarks and parity tests run on these random-init weights (bench.py says so).
"""
from __future__ import annotations

import math
import os
from typing import Dict

import torch

from .config import SynthConfig

StateDict = Dict[str, torch.Tensor]


class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)
        self.sd: StateDict = {}

    def linear(self, name, n_out, n_in, bias=True, gain=1.0):
        self.sd[name + ".weight"] = torch.randn(n_out, n_in, generator=self.g) * (gain / math.sqrt(n_in))
        if bias:
            self.sd[name + ".bias"] = torch.randn(n_out, generator=self.g) * 0.02

    def conv(self, name, c_out, c_in, k, bias=True, gain=1.0):
        self.sd[name + ".weight"] = torch.randn(c_out, c_in, k, generator=self.g) * (gain / math.sqrt(c_in * k))
        if bias:
            self.sd[name + ".bias"] = torch.randn(c_out, generator=self.g) * 0.02

    def convT(self, name, c_in, c_out, k, stride):
        self.sd[name + ".weight"] = torch.randn(c_in, c_out, k, generator=self.g) * math.sqrt(stride / (c_in * k))
        self.sd[name + ".bias"] = torch.randn(c_out, generator=self.g) * 0.02

    def norm(self, name, c):
        self.sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=self.g)
        self.sd[name + ".bias"] = 0.05 * torch.randn(c, generator=self.g)

    def emb(self, name, n, c, scale=1.0):
        self.sd[name + ".weight"] = torch.randn(n, c, generator=self.g) * scale

    def tensor(self, name, *shape, scale=1.0, offset=0.0):
        self.sd[name] = torch.randn(*shape, generator=self.g) * scale + offset


class _Shapes(_Init):
    """The same walk over the architecture with shapes instead of tensors: the expected-key manifest of a checkpoint."""

    def __init__(self):
        self.sd = {}

    def linear(self, name, n_out, n_in, bias=True, gain=1.0):
        self.sd[name + ".weight"] = (n_out, n_in)
        if bias:
            self.sd[name + ".bias"] = (n_out,)

    def conv(self, name, c_out, c_in, k, bias=True, gain=1.0):
        self.sd[name + ".weight"] = (c_out, c_in, k)
        if bias:
            self.sd[name + ".bias"] = (c_out,)

    def convT(self, name, c_in, c_out, k, stride):
        self.sd[name + ".weight"] = (c_in, c_out, k)
        self.sd[name + ".bias"] = (c_out,)

    def norm(self, name, c):
        self.sd[name + ".weight"] = (c,)
        self.sd[name + ".bias"] = (c,)

    def emb(self, name, n, c, scale=1.0):
        self.sd[name + ".weight"] = (n, c)

    def tensor(self, name, *shape, scale=1.0, offset=0.0):
        self.sd[name] = tuple(shape)


def _relpos_encoder(I: _Init, p: str, in_dim: int, d: int, heads: int, ffn: int, layers: int, conformer: bool):
    I.linear(f"{p}.embed.out.0", d, in_dim)
    I.norm(f"{p}.embed.out.1", d)
    n1, n2 = ("norm_mha", "norm_ff") if conformer else ("norm1", "norm2")
    for i in range(layers):
        q = f"{p}.encoders.{i}"
        I.norm(f"{q}.{n1}", d)
        I.norm(f"{q}.{n2}", d)
        for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
            I.linear(f"{q}.self_attn.{nm}", d, d)
        I.linear(f"{q}.self_attn.linear_pos", d, d, bias=False)
        I.tensor(f"{q}.self_attn.pos_bias_u", heads, d // heads, scale=0.1)
        I.tensor(f"{q}.self_attn.pos_bias_v", heads, d // heads, scale=0.1)
        I.linear(f"{q}.feed_forward.w_1", ffn, d)
        I.linear(f"{q}.feed_forward.w_2", d, ffn)
    I.norm(f"{p}.after_norm", d)


def make_lm_weights(cfg: SynthConfig, seed: int = 0, I: "_Init" = None) -> StateDict:
    I = I or _Init(seed)
    I.emb("text_embedding", cfg.text_vocab, cfg.text_dim)
    _relpos_encoder(I, "text_encoder", cfg.text_dim, cfg.lm_dim, cfg.lm_heads, cfg.lm_ffn, cfg.lm_text_layers, True)
    I.linear("text_encoder_affine_layer", cfg.lm_dim, cfg.lm_dim)
    I.emb("llm_embedding", 2, cfg.lm_dim)
    I.emb("speech_embedding", cfg.speech_vocab, cfg.lm_dim)
    I.linear("spk_embed_affine_layer", cfg.lm_dim, cfg.spk_dim)
    _relpos_encoder(I, "llm", cfg.lm_dim, cfg.lm_dim, cfg.lm_heads, cfg.lm_ffn, cfg.lm_layers, False)
    I.linear("llm_decoder", cfg.speech_vocab + 1, cfg.lm_dim, gain=2.0)
    return I.sd


def _resnet(I: _Init, p: str, c_in: int, c_out: int, t_dim: int):
    I.conv(f"{p}.block1.block.0", c_out, c_in, 3)
    I.norm(f"{p}.block1.block.1", c_out)
    I.linear(f"{p}.mlp.1", c_out, t_dim)
    I.conv(f"{p}.block2.block.0", c_out, c_out, 3)
    I.norm(f"{p}.block2.block.1", c_out)
    I.conv(f"{p}.res_conv", c_out, c_in, 1)


def _tfm(I: _Init, p: str, c: int, heads: int):
    inner = heads * 64
    I.norm(f"{p}.norm1", c)
    for nm in ("to_q", "to_k", "to_v"):
        I.linear(f"{p}.attn1.{nm}", inner, c, bias=False)
    I.linear(f"{p}.attn1.to_out.0", c, inner)
    I.norm(f"{p}.norm3", c)
    I.linear(f"{p}.ff.net.0.proj", 4 * c, c)
    I.linear(f"{p}.ff.net.2", c, 4 * c)


def make_flow_weights(cfg: SynthConfig, seed: int = 1, I: "_Init" = None) -> StateDict:
    I = I or _Init(seed)
    I.emb("input_embedding", cfg.speech_vocab, cfg.flow_dim)
    I.linear("spk_embed_affine_layer", cfg.mel, cfg.spk_dim)
    _relpos_encoder(I, "encoder", cfg.flow_dim, cfg.flow_dim, cfg.flow_heads, cfg.flow_ffn, cfg.flow_layers, True)
    I.linear("encoder_proj", cfg.mel, cfg.flow_dim)
    for j in range(4):
        I.conv(f"length_regulator.model.{3 * j}", cfg.mel, cfg.mel, 3)
        I.norm(f"length_regulator.model.{3 * j + 1}", cfg.mel)
    I.conv("length_regulator.model.12", cfg.mel, cfg.mel, 1)
    e = "decoder.estimator"
    td = cfg.est_time_dim
    I.linear(f"{e}.time_mlp.linear_1", td, cfg.est_in)
    I.linear(f"{e}.time_mlp.linear_2", td, td)
    ch = cfg.est_channels
    c_prev = cfg.est_in
    for i, c in enumerate(ch):
        _resnet(I, f"{e}.down_blocks.{i}.0", c_prev, c, td)
        for j in range(cfg.est_tfm_per_block):
            _tfm(I, f"{e}.down_blocks.{i}.1.{j}", c, cfg.est_heads)
        last = i == len(ch) - 1
        I.conv(f"{e}.down_blocks.{i}.2" + ("" if last else ".conv"), c, c, 3)
        c_prev = c
    for i in range(cfg.est_mid_blocks):
        _resnet(I, f"{e}.mid_blocks.{i}.0", ch[-1], ch[-1], td)
        for j in range(cfg.est_tfm_per_block):
            _tfm(I, f"{e}.mid_blocks.{i}.1.{j}", ch[-1], cfg.est_heads)
    up = tuple(ch[::-1]) + (ch[0],)
    for i in range(len(up) - 1):
        _resnet(I, f"{e}.up_blocks.{i}.0", 2 * up[i], up[i + 1], td)
        for j in range(cfg.est_tfm_per_block):
            _tfm(I, f"{e}.up_blocks.{i}.1.{j}", up[i + 1], cfg.est_heads)
        last = i == len(up) - 2
        if last:
            I.conv(f"{e}.up_blocks.{i}.2", up[i + 1], up[i + 1], 3)
        else:
            I.convT(f"{e}.up_blocks.{i}.2.conv", up[i + 1], up[i + 1], 4, 2)
    I.conv(f"{e}.final_block.block.0", up[-1], up[-1], 3)
    I.norm(f"{e}.final_block.block.1", up[-1])
    I.conv(f"{e}.final_proj", cfg.mel, up[-1], 1)
    return I.sd


def make_hift_weights(cfg: SynthConfig, seed: int = 2, I: "_Init" = None) -> StateDict:
    I = I or _Init(seed)
    c_prev = cfg.mel
    for j in range(5):
        I.conv(f"f0_predictor.condnet.{2 * j}", cfg.f0_channels, c_prev, 3)
        c_prev = cfg.f0_channels
    I.linear("f0_predictor.classifier", 1, cfg.f0_channels, gain=4.0)
    if not isinstance(I, _Shapes):
        I.sd["f0_predictor.classifier.bias"] = torch.tensor([120.0])  # voiced-range f0 with random weights
    I.linear("m_source.l_linear", 1, cfg.nb_harmonics + 1, gain=3.0)
    base = cfg.hift_base
    I.conv("conv_pre", base, cfg.mel, 7)
    n_up = len(cfg.up_rates)
    # cumulative downsampling of the source STFT towards each upsampling stage
    down = [1]
    for r in cfg.up_rates[::-1][:-1]:
        down.append(down[-1] * r)
    down = down[::-1]
    for i, r in enumerate(cfg.up_rates):
        c_out = base // (2 ** (i + 1))
        I.convT(f"ups.{i}", base // (2 ** i), c_out, 2 * r, r)
        u = down[i]
        if u == 1:
            I.conv(f"source_downs.{i}", c_out, 18, 1)
        else:
            I.conv(f"source_downs.{i}", c_out, 18, 2 * u)
        k = cfg.src_res_kernels[i]
        for j in range(len(cfg.res_dils)):
            I.conv(f"source_resblocks.{i}.convs1.{j}", c_out, c_out, k, gain=0.5)
            I.conv(f"source_resblocks.{i}.convs2.{j}", c_out, c_out, k, gain=0.5)
            I.tensor(f"source_resblocks.{i}.activations1.{j}.alpha", c_out, scale=0.1, offset=1.0)
            I.tensor(f"source_resblocks.{i}.activations2.{j}.alpha", c_out, scale=0.1, offset=1.0)
        for kk, k in enumerate(cfg.res_kernels):
            n = i * len(cfg.res_kernels) + kk
            for j in range(len(cfg.res_dils)):
                I.conv(f"resblocks.{n}.convs1.{j}", c_out, c_out, k, gain=0.5)
                I.conv(f"resblocks.{n}.convs2.{j}", c_out, c_out, k, gain=0.5)
                I.tensor(f"resblocks.{n}.activations1.{j}.alpha", c_out, scale=0.1, offset=1.0)
                I.tensor(f"resblocks.{n}.activations2.{j}.alpha", c_out, scale=0.1, offset=1.0)
    I.conv("conv_post", 18, base // (2 ** n_up), 7, gain=0.3)
    return I.sd


def make_all(cfg: SynthConfig, seed: int = 0) -> Dict[str, StateDict]:
    return {"llm": make_lm_weights(cfg, seed), "flow": make_flow_weights(cfg, seed + 1),
            "hift": make_hift_weights(cfg, seed + 2)}


def _fold_weight_norm(sd: StateDict) -> StateDict:
    """torch weight_norm checkpoints store (weight_g, weight_v) or parametrizations.*: fold to .weight."""
    out = dict(sd)
    for k in list(sd):
        for g_sfx, v_sfx in ((".weight_g", ".weight_v"), (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
            if k.endswith(g_sfx):
                base = k[: -len(g_sfx)]
                g, v = sd[k], sd[base + v_sfx]
                norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
                out[base + ".weight"] = v * (g / norm)
                out.pop(k, None)
                out.pop(base + v_sfx, None)
    return out


def expected_shapes(cfg: SynthConfig) -> Dict[str, Dict[str, tuple]]:
    """{"llm" | "flow" | "hift": {state-dict key: shape}} -- every tensor the engine reads, derived from ``cfg`` by the same walk over
    the architecture that builds the synthetic weights (so the two cannot drift apart)."""
    return {"llm": make_lm_weights(cfg, I=_Shapes()), "flow": make_flow_weights(cfg, I=_Shapes()), "hift": make_hift_weights(cfg, I=_Shapes())}


# non-parameter entries real checkpoints are known to carry ([EXT]-recalled: registered buffers of upstream modules); never read here
IGNORABLE_KEY_PATTERNS = (r"\.num_batches_tracked$", r"(^|\.)stft_window$", r"\.pos_enc\.pe$", r"^m_source\.l_sin_gen\.", r"^f0_upsamp\.")


def check_state_dicts(state: Dict[str, StateDict], cfg: SynthConfig, strict: bool = False) -> None:
    """Hold loaded state dicts (weight norm already folded) to the manifest of ``cfg``: missing and mis-shaped tensors are reported
    TOGETHER in one ValueError (a wrong checkpoint / config pair otherwise surfaces as a KeyError deep inside engine construction);
    unexpected keys are listed in a warning (``strict``: in the error) unless they match ``IGNORABLE_KEY_PATTERNS``."""
    import re
    import warnings

    want = expected_shapes(cfg)
    problems, extra = [], []
    for part in ("llm", "flow", "hift"):
        sd = state.get(part)
        if sd is None:
            problems.append(f"{part}: no state dict")
            continue
        missing = sorted(k for k in want[part] if k not in sd)
        wrong = sorted((k, tuple(sd[k].shape), want[part][k]) for k in want[part] if k in sd and tuple(sd[k].shape) != tuple(want[part][k]))
        unexpected = sorted(k for k in sd if k not in want[part] and not any(re.search(p, k) for p in IGNORABLE_KEY_PATTERNS))
        if missing:
            problems.append(f"{part}.pt: {len(missing)} missing key(s): " + ", ".join(missing[:12]) + (" ..." if len(missing) > 12 else ""))
        if wrong:
            problems.append(f"{part}.pt: {len(wrong)} mis-shaped tensor(s): " +
                            ", ".join(f"{k} is {got}, expected {exp}" for k, got, exp in wrong[:12]) + (" ..." if len(wrong) > 12 else ""))
        if unexpected:
            extra.append(f"{part}.pt: {len(unexpected)} unexpected key(s): " + ", ".join(unexpected[:12]) + (" ..." if len(unexpected) > 12 else ""))
    if strict:
        problems += extra
    if problems:
        raise ValueError("checkpoint does not match the model config (SynthConfig / model_dir/astts.json):\n  " + "\n  ".join(problems + ([] if strict else extra)))
    if extra:
        warnings.warn("checkpoint carries tensors the engine does not read:\n  " + "\n  ".join(extra), RuntimeWarning, stacklevel=2)


def load_state_dicts(model_dir: str, cfg: SynthConfig = None, strict: bool = False) -> Dict[str, StateDict]:
    """Real checkpoints, when a CosyVoice-300M directory is supplied at run time (none exists here): ``llm.pt`` / ``flow.pt`` /
    ``hift.pt`` as upstream saves them (weight_norm in either of torch's two forms is folded).  With ``cfg`` the result is held to
    the config's key / shape manifest (``check_state_dicts``)."""
    out = {}
    for name in ("llm", "flow", "hift"):
        path = os.path.join(model_dir, f"{name}.pt")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        sd = torch.load(path, map_location="cpu", weights_only=True)
        out[name] = _fold_weight_norm({k: v.float() for k, v in sd.items() if torch.is_tensor(v)})
    if cfg is not None:
        check_state_dicts(out, cfg, strict)
    return out
