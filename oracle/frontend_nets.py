"""CPU oracle for the learned half of the frontend: fp32 PyTorch restatement.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py for who may import this).

PARITY UNPINNED against the reference, block-pinned against third-party code.  The reference runs these two networks as ONNX
files through onnxruntime inside ``CosyVoice(model_dir)`` (/root/reference/tts_with_rag.py:159) on every prompt it loads
(/root/reference/tts_with_rag.py:179-195, /root/reference/tts_with_style_and_timbre.py:83-93); neither the files nor
onnxruntime exist here.  What is restated is the PUBLISHED architecture of each ([EXT]-recalled):

* ``speech_tokens``: CosyVoice's supervised semantic tokenizer v1 as re-implemented publicly in s3tokenizer
  (``S3Tokenizer("speech_tokenizer_v1")``): a Whisper audio encoder -- conv1d(128 -> d, k 3, pad 1) + GELU, conv1d(d -> d,
  k 3, stride 2, pad 1) + GELU, + sinusoidal positions, ``layers`` pre-norm blocks
  ``x += attn(ln(x)); x += mlp(ln(x))`` with Whisper's attention (query / value / out with bias, key without; q and k each
  scaled by head_dim^-0.25; key padding mask) and a 4x GELU MLP, NO final LayerNorm -- then a Euclidean codebook:
  ``argmax -(|x|^2 - 2 x.e + |e|^2)`` of the L2-normalised frame.  The encoder BLOCK is pinned against transformers'
  ``WhisperEncoderLayer`` and the conv stem + positions against ``WhisperEncoder`` (tests/golden/make_frontend_fixtures.py ->
  tests/test_oracle_frontend_nets.py).
* ``speaker_embedding``: 3D-Speaker's CAM++ (``CAMPPlus``): FCM head (conv2d 1 -> 32 + BN + ReLU, two stages of two
  BasicResBlocks with stride 2 along frequency, conv2d stride (2, 1) + BN + ReLU, flattened to 32 * feat_dim / 8 channels),
  TDNN (k 5, stride 2) + BN + ReLU, three densely connected blocks of CAM layers (BN-ReLU -> 1x1 -> BN-ReLU -> dilated k 3
  "local" conv gated by sigmoid(linear2(relu(linear1(mean over time + 100-frame segment mean))))), transit layers halving the
  channels, BN-ReLU, statistics pooling (mean | unbiased std), dense 192 + BN without affine.  Eval-mode BatchNorm.  No
  third-party twin of CAM++ exists in this environment: its wiring stays [EXT]-recalled; its building blocks (eval BatchNorm,
  convolutions, avg_pool1d(ceil_mode), std) are torch's own.

Both are the checkers for the HIP path (astts/frontend_nets.py) under identical seeded weights
(astts.frontend_weights.make_*_weights).  Tensors: mel ``[B, n_mels, T]`` (as whisper_log_mel returns it), fbank ``[B, T, 80]``.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------ speech tokenizer
def whisper_block(sd: SD, p: str, x: torch.Tensor, heads: int, key_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One ResidualAttentionBlock.  x [B, T, d]; key_mask bool [B, T] (True = valid key)."""
    b, t, d = x.shape
    hd = d // heads
    h = F.layer_norm(x, (d,), sd[p + "attn_ln.weight"], sd[p + "attn_ln.bias"], 1e-5)
    q = F.linear(h, sd[p + "attn.query.weight"], sd[p + "attn.query.bias"])
    k = F.linear(h, sd[p + "attn.key.weight"])
    v = F.linear(h, sd[p + "attn.value.weight"], sd[p + "attn.value.bias"])
    scale = hd ** -0.25
    q = q.view(b, t, heads, hd).permute(0, 2, 1, 3) * scale
    k = k.view(b, t, heads, hd).permute(0, 2, 3, 1) * scale
    v = v.view(b, t, heads, hd).permute(0, 2, 1, 3)
    qk = q @ k
    if key_mask is not None:
        qk = qk.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    w = torch.softmax(qk.float(), dim=-1)
    a = (w @ v).permute(0, 2, 1, 3).reshape(b, t, d)
    x = x + F.linear(a, sd[p + "attn.out.weight"], sd[p + "attn.out.bias"])
    h = F.layer_norm(x, (d,), sd[p + "mlp_ln.weight"], sd[p + "mlp_ln.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    return x + F.linear(h, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])


def tokenizer_stem(sd: SD, mel: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """mel [B, n_mels, T] -> [B, ceil(T / 2), d]: the two convolutions + positions."""
    x = F.gelu(F.conv1d(mel, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    return x + positions[: x.shape[1]]


def tokenizer_encode(sd: SD, cfg, mel: torch.Tensor, mel_lens: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (frames [B, T', d], lens [B]) with T' = (T - 1) // 2 + 1."""
    from astts.frontend_weights import sinusoids

    b, _, t = mel.shape
    if mel_lens is None:
        mel_lens = torch.full((b,), t, dtype=torch.int64)
    x = tokenizer_stem(sd, mel, sinusoids(cfg.n_ctx, cfg.d))
    lens = (mel_lens + 2 - 2 - 1) // 2 + 1                # conv2: k 3, stride 2, pad 1 (conv1 keeps the length)
    mask = torch.arange(x.shape[1])[None, :] < lens[:, None]
    for i in range(cfg.layers):
        x = whisper_block(sd, f"encoder.blocks.{i}.", x, cfg.heads, mask)
    return x, lens


def vq_encode(sd: SD, cfg, frames: torch.Tensor) -> torch.Tensor:
    """frames [..., d] -> int64 codes: nearest codebook entry (squared distance, first index on ties) of the normalised frame."""
    x = frames.to(torch.float64)
    if cfg.normalize:
        x = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    e = sd["quantizer._codebook.embed"].to(torch.float64)
    flat = x.reshape(-1, x.shape[-1])
    d2 = ((flat[:, None, :] - e[None, :, :]) ** 2).sum(-1) if flat.shape[0] * e.shape[0] * e.shape[1] <= (1 << 27) else \
        torch.stack([((e - row) ** 2).sum(-1) for row in flat])
    return d2.argmin(dim=-1).reshape(x.shape[:-1])


def speech_tokens(sd: SD, cfg, mel: torch.Tensor, mel_lens: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mel [B, n_mels, T] -> (codes int64 [B, T'], lens); codes behind a row's length are unspecified."""
    x, lens = tokenizer_encode(sd, cfg, mel, mel_lens)
    return vq_encode(sd, cfg, x), lens


# ------------------------------------------------------------------------------------------ CAM++
def _bn(sd: SD, p: str, x: torch.Tensor, eps: float, relu: bool = True) -> torch.Tensor:
    y = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd.get(p + ".weight"), sd.get(p + ".bias"), False, 0.0, eps)
    return F.relu(y) if relu else y


def _res_block(sd: SD, p: str, x: torch.Tensor, stride: int, eps: float) -> torch.Tensor:
    out = _bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=(stride, 1), padding=1), eps)
    out = _bn(sd, p + "bn2", F.conv2d(out, sd[p + "conv2.weight"], padding=1), eps, relu=False)
    if (p + "shortcut.0.weight") in sd:
        x = _bn(sd, p + "shortcut.1", F.conv2d(x, sd[p + "shortcut.0.weight"], stride=(stride, 1)), eps, relu=False)
    return F.relu(out + x)


def campplus_head(sd: SD, cfg, fbank: torch.Tensor) -> torch.Tensor:
    """fbank [B, T, F] -> [B, 32 * F / 8, T]"""
    eps = cfg.bn_eps
    x = fbank.permute(0, 2, 1).unsqueeze(1)                       # [B, 1, F, T]
    out = _bn(sd, "head.bn1", F.conv2d(x, sd["head.conv1.weight"], padding=1), eps)
    for li in (1, 2):
        out = _res_block(sd, f"head.layer{li}.0.", out, 2, eps)
        out = _res_block(sd, f"head.layer{li}.1.", out, 1, eps)
    out = _bn(sd, "head.bn2", F.conv2d(out, sd["head.conv2.weight"], stride=(2, 1), padding=1), eps)
    return out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])


def _seg_pool(x: torch.Tensor, seg_len: int) -> torch.Tensor:
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    seg = seg.unsqueeze(-1).expand(*seg.shape, seg_len).reshape(*seg.shape[:-1], -1)
    return seg[..., : x.shape[-1]]


def cam_dense_layer(sd: SD, p: str, x: torch.Tensor, k: int, dil: int, seg_len: int, eps: float) -> torch.Tensor:
    """x [B, C, T] -> the layer's 32 new channels [B, growth, T]"""
    h = F.conv1d(_bn(sd, p + "nonlinear1.batchnorm", x, eps), sd[p + "linear1.weight"])
    h = _bn(sd, p + "nonlinear2.batchnorm", h, eps)
    y = F.conv1d(h, sd[p + "cam_layer.linear_local.weight"], padding=(k - 1) // 2 * dil, dilation=dil)
    ctx = h.mean(-1, keepdim=True) + _seg_pool(h, seg_len)
    ctx = F.relu(F.conv1d(ctx, sd[p + "cam_layer.linear1.weight"], sd[p + "cam_layer.linear1.bias"]))
    m = torch.sigmoid(F.conv1d(ctx, sd[p + "cam_layer.linear2.weight"], sd[p + "cam_layer.linear2.bias"]))
    return y * m


def campplus_xvector(sd: SD, cfg, x: torch.Tensor, return_frames: bool = False) -> torch.Tensor:
    """x [B, head_out, T] -> embedding [B, emb]"""
    eps = cfg.bn_eps
    x = _bn(sd, "xvector.tdnn.nonlinear.batchnorm", F.conv1d(x, sd["xvector.tdnn.linear.weight"], stride=2, padding=2), eps)
    for bi, (layers, k, dil) in enumerate(cfg.blocks):
        for li in range(layers):
            x = torch.cat([x, cam_dense_layer(sd, f"xvector.block{bi + 1}.tdnnd{li + 1}.", x, k, dil, cfg.seg_len, eps)], dim=1)
        p = f"xvector.transit{bi + 1}."
        x = F.conv1d(_bn(sd, p + "nonlinear.batchnorm", x, eps), sd[p + "linear.weight"])
    x = _bn(sd, "xvector.out_nonlinear.batchnorm", x, eps)
    if return_frames:
        return x
    stats = torch.cat([x.mean(-1), x.std(-1, unbiased=True)], dim=-1)
    e = F.conv1d(stats.unsqueeze(-1), sd["xvector.dense.linear.weight"]).squeeze(-1)
    return _bn(sd, "xvector.dense.nonlinear.batchnorm", e, eps, relu=False)


def speaker_embedding(sd: SD, cfg, fbank: torch.Tensor) -> torch.Tensor:
    """fbank [B, T, feat_dim] (mean over time already removed, as upstream's frontend does) -> [B, emb]"""
    return campplus_xvector(sd, cfg, campplus_head(sd, cfg, fbank))
