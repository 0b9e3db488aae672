"""Shapes and state dicts of the two learned frontend networks.

The reference loads them as ONNX files inside ``CosyVoice(model_dir)`` (/root/reference/tts_with_rag.py:159) [EXT]:
  speech_tokenizer_v1.onnx  a Whisper-style audio encoder (two convolutions, the second with stride 2 -> 50 frames/s; sinusoidal
                            positions; 6 pre-norm transformer blocks, d 1280, 20 heads x 64, FFN 5120 GELU) over the 128-bin Whisper
                            log-mel of the 16 kHz prompt, followed by a 4096-entry Euclidean codebook: token = arg-min distance
                            of the L2-normalised frame ("supervised semantic tokens", CosyVoice paper; the public re-implementation
                            is s3tokenizer's S3Tokenizer v1, whose parameter names the state dict here follows)
  campplus.onnx             3D-Speaker's CAM++ (FCM 2-D convolution head on the 80-bin Kaldi fbank, a TDNN layer, three
                            densely connected TDNN blocks of 12 / 24 / 16 context-aware-masking layers with transit layers,
                            statistics pooling, a 192-d dense layer); parameter names follow 3D-Speaker's ``CAMPPlus`` module
No trained weights exist offline, so -- like every other stage of this build -- the networks are exercised on SEEDED SYNTHETIC
weights at the published shapes; real ones load through the same names (``load_frontend_weights``: a ``.pt`` state dict or the
ONNX file's initializers).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

StateDict = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class SpeechTokenizerShape:
    n_mels: int = 128
    n_ctx: int = 1500            # 30 s at 50 frames/s (upstream refuses longer prompts)
    d: int = 1280
    heads: int = 20
    layers: int = 6
    codes: int = 4096
    normalize: bool = True       # frames are L2-normalised in front of the codebook (s3tokenizer VectorQuantization.encode)

    @staticmethod
    def tiny() -> "SpeechTokenizerShape":
        return SpeechTokenizerShape(n_ctx=400, d=128, heads=2, layers=2, codes=64)


@dataclass(frozen=True)
class CamPlusShape:
    feat_dim: int = 80
    emb: int = 192
    growth: int = 32
    bn_size: int = 4
    init_channels: int = 128
    m_channels: int = 32         # FCM head
    blocks: Tuple[Tuple[int, int, int], ...] = ((12, 3, 1), (24, 3, 2), (16, 3, 2))     # (layers, kernel, dilation)
    seg_len: int = 100
    bn_eps: float = 1e-5

    @staticmethod
    def tiny() -> "CamPlusShape":
        return CamPlusShape(feat_dim=16, emb=24, blocks=((2, 3, 1), (3, 3, 2)), init_channels=64, seg_len=20)

    @property
    def head_out(self) -> int:
        return self.m_channels * (self.feat_dim // 8)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Whisper's fixed positions: [sin | cos] over log-spaced timescales."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def _drawer(seed):
    """``r(*shape, std=)``: seeded Gaussians -- or, for ``seed`` None, meta tensors (shapes only: the manifest of a network)."""
    if seed is None:
        return lambda *s, std: torch.empty(*s, device="meta")
    g = torch.Generator().manual_seed(seed)
    return lambda *s, std: torch.randn(*s, generator=g) * std


def make_speech_tokenizer_weights(cfg: SpeechTokenizerShape, seed: Optional[int] = 0) -> StateDict:
    r = _drawer(seed)
    d = cfg.d
    sd: StateDict = {
        "encoder.conv1.weight": r(d, cfg.n_mels, 3, std=1.0 / math.sqrt(3 * cfg.n_mels)), "encoder.conv1.bias": r(d, std=0.1),
        "encoder.conv2.weight": r(d, d, 3, std=1.5 / math.sqrt(3 * d)), "encoder.conv2.bias": r(d, std=0.1),
    }
    for i in range(cfg.layers):
        p = f"encoder.blocks.{i}."
        sd[p + "attn_ln.weight"] = 1.0 + r(d, std=0.1)
        sd[p + "attn_ln.bias"] = r(d, std=0.1)
        sd[p + "attn.query.weight"] = r(d, d, std=1.0 / math.sqrt(d))
        sd[p + "attn.query.bias"] = r(d, std=0.1)
        sd[p + "attn.key.weight"] = r(d, d, std=1.0 / math.sqrt(d))          # (Whisper's key projection has no bias)
        sd[p + "attn.value.weight"] = r(d, d, std=1.0 / math.sqrt(d))
        sd[p + "attn.value.bias"] = r(d, std=0.1)
        sd[p + "attn.out.weight"] = r(d, d, std=0.5 / math.sqrt(d))
        sd[p + "attn.out.bias"] = r(d, std=0.05)
        sd[p + "mlp_ln.weight"] = 1.0 + r(d, std=0.1)
        sd[p + "mlp_ln.bias"] = r(d, std=0.1)
        sd[p + "mlp.0.weight"] = r(4 * d, d, std=1.0 / math.sqrt(d))
        sd[p + "mlp.0.bias"] = r(4 * d, std=0.1)
        sd[p + "mlp.2.weight"] = r(d, 4 * d, std=0.5 / math.sqrt(4 * d))
        sd[p + "mlp.2.bias"] = r(d, std=0.05)
    code = r(cfg.codes, d, std=1.0)
    sd["quantizer._codebook.embed"] = code / code.norm(dim=1, keepdim=True) if cfg.normalize else code
    return sd


def _bn(sd: StateDict, name: str, c: int, r, affine: bool = True) -> None:
    if affine:
        sd[name + ".weight"] = 1.0 + r(c, std=0.2)
        sd[name + ".bias"] = r(c, std=0.2)
    sd[name + ".running_mean"] = r(c, std=0.3)
    sd[name + ".running_var"] = (1.0 + r(c, std=0.3)).abs() + 0.2


def make_campplus_weights(cfg: CamPlusShape, seed: Optional[int] = 0) -> StateDict:
    r = _drawer(seed)
    m = cfg.m_channels
    sd: StateDict = {"head.conv1.weight": r(m, 1, 3, 3, std=1.0 / 3.0)}
    _bn(sd, "head.bn1", m, r)
    for li in (1, 2):
        for bi in (0, 1):
            p = f"head.layer{li}.{bi}."
            sd[p + "conv1.weight"] = r(m, m, 3, 3, std=1.4 / math.sqrt(9 * m))
            _bn(sd, p + "bn1", m, r)
            sd[p + "conv2.weight"] = r(m, m, 3, 3, std=1.0 / math.sqrt(9 * m))
            _bn(sd, p + "bn2", m, r)
            if bi == 0:                          # stride 2 along frequency: projection shortcut
                sd[p + "shortcut.0.weight"] = r(m, m, 1, 1, std=1.0 / math.sqrt(m))
                _bn(sd, p + "shortcut.1", m, r)
    sd["head.conv2.weight"] = r(m, m, 3, 3, std=1.4 / math.sqrt(9 * m))
    _bn(sd, "head.bn2", m, r)
    ch = cfg.init_channels
    sd["xvector.tdnn.linear.weight"] = r(ch, cfg.head_out, 5, std=1.4 / math.sqrt(5 * cfg.head_out))
    _bn(sd, "xvector.tdnn.nonlinear.batchnorm", ch, r)
    bnc = cfg.bn_size * cfg.growth
    for bi, (layers, k, _dil) in enumerate(cfg.blocks):
        for li in range(layers):
            p = f"xvector.block{bi + 1}.tdnnd{li + 1}."
            cin = ch + li * cfg.growth
            _bn(sd, p + "nonlinear1.batchnorm", cin, r)
            sd[p + "linear1.weight"] = r(bnc, cin, 1, std=1.4 / math.sqrt(cin))
            _bn(sd, p + "nonlinear2.batchnorm", bnc, r)
            sd[p + "cam_layer.linear_local.weight"] = r(cfg.growth, bnc, k, std=1.4 / math.sqrt(k * bnc))
            sd[p + "cam_layer.linear1.weight"] = r(bnc // 2, bnc, 1, std=1.4 / math.sqrt(bnc))
            sd[p + "cam_layer.linear1.bias"] = r(bnc // 2, std=0.2)
            sd[p + "cam_layer.linear2.weight"] = r(cfg.growth, bnc // 2, 1, std=2.0 / math.sqrt(bnc // 2))
            sd[p + "cam_layer.linear2.bias"] = r(cfg.growth, std=0.2)
        ch = ch + layers * cfg.growth
        p = f"xvector.transit{bi + 1}."
        _bn(sd, p + "nonlinear.batchnorm", ch, r)
        sd[p + "linear.weight"] = r(ch // 2, ch, 1, std=1.4 / math.sqrt(ch))
        ch //= 2
    _bn(sd, "xvector.out_nonlinear.batchnorm", ch, r)
    sd["xvector.dense.linear.weight"] = r(cfg.emb, 2 * ch, 1, std=1.0 / math.sqrt(2 * ch))
    _bn(sd, "xvector.dense.nonlinear.batchnorm", cfg.emb, r, affine=False)
    return sd


def manifest(sd: StateDict) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in sd.items()}


def manifest_for(maker, shape) -> Dict[str, Tuple[int, ...]]:
    """name -> shape of a network at ``shape`` without drawing its weights (``maker(shape, None)`` runs on meta tensors)."""
    return manifest(maker(shape, None))


def check_against_manifest(sd: StateDict, want: Dict[str, Tuple[int, ...]], what: str) -> StateDict:
    """One readable error for every missing or mis-shaped tensor (extra tensors are ignored: ``num_batches_tracked`` etc.)."""
    problems = []
    for k, shp in want.items():
        if k not in sd:
            problems.append(f"missing {k} {list(shp)}")
        elif tuple(sd[k].shape) != shp:
            problems.append(f"{k}: shape {list(sd[k].shape)}, expected {list(shp)}")
    if problems:
        head = "; ".join(problems[:8]) + (f"; ... {len(problems) - 8} more" if len(problems) > 8 else "")
        have = ", ".join(list(sd)[:6])
        raise ValueError(f"{what}: {len(problems)} of {len(want)} tensors do not match the network's manifest: {head}.  "
                         f"The file holds {len(sd)} tensors (first names: {have}).")
    return {k: sd[k].to(torch.float32) for k in want}


def load_frontend_weights(model_dir: str, stem: str, want: Dict[str, Tuple[int, ...]]) -> Optional[StateDict]:
    """``model_dir/<stem>.pt`` (a state dict) or ``model_dir/<stem>.onnx`` (graph initializers under the same names) held to
    the manifest; None when neither file exists."""
    pt, onnx = os.path.join(model_dir, stem + ".pt"), os.path.join(model_dir, stem + ".onnx")
    if os.path.exists(pt):
        sd = torch.load(pt, map_location="cpu", weights_only=True)
        return check_against_manifest(sd, want, pt)
    if os.path.exists(onnx):
        from .onnx_weights import read_initializers

        sd = {k: torch.from_numpy(v) for k, v in read_initializers(onnx).items()}
        return check_against_manifest(sd, want, onnx)
    return None
