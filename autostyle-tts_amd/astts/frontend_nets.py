"""The learned half of the frontend on the GPU: the speech tokenizer and the speaker-embedding
network that ``CosyVoice(model_dir)`` loads as ONNX files (/root/reference/tts_with_rag.py:159) and runs through onnxruntime
on every prompt it is given (/root/reference/tts_with_rag.py:179-195, /root/reference/tts_with_style_and_timbre.py:83-93).

Here both run as HIP kernels behind the C ABI -- no onnxruntime, no torch arithmetic:
  SpeechTokenizerV1   128-bin Whisper log-mel (astts_op_whisper_log_mel) -> two convolutions (astts_op_gemm_ex, taps 3, GELU
                      epilogue; the positions ride on the second one's residual operand) -> ``layers`` pre-norm blocks
                      (astts_op_layernorm -> fused q | k | v projection -> astts_op_attn_mha_ex at head dimension 64 ->
                      out-projection + residual -> LayerNorm -> FFN-in + exact GELU (fp16) -> FFN-out + residual) ->
                      astts_op_l2_normalize -> arg-min over the 4096-entry codebook = a k = 1 search of astts_knn_* under
                      ASTTS_METRIC_L2 (fp64-certified: the id is the exact arg-min of the frame this path computed)
  CamPlusSpeakerNet   80-bin Kaldi fbank (astts_op_kaldi_fbank) minus its mean over time -> FCM head (3 x 3 convolutions as
                      3-tap GEMMs over time on astts_op_freq_unfold's frequency windows, eval BatchNorm folded into the
                      weights) -> TDNN -> three dense blocks of context-aware-masking layers (astts_op_affine_act,
                      astts_op_cam_context, astts_op_cam_gate around 1x1 / dilated GEMMs, every layer writing its 32 new
                      channels into the block's concatenation buffer in place) -> astts_op_stats_pool -> dense 192
Definition and tolerances: oracle/frontend_nets.py, tests/test_frontend_nets_gpu.py.  Weights: astts/frontend_weights.py
(seeded synthetic at the published shapes, or ``speech_tokenizer_v1.{pt,onnx}`` / ``campplus.{pt,onnx}`` from model_dir).
torch is used for HBM buffers, views and layout transposes only.
"""
from __future__ import annotations

from ctypes import c_float, c_int32, c_int64, c_void_p
from typing import Dict, Optional

import torch

from . import _lib, audio, ops
from .frontend_weights import CamPlusShape, SpeechTokenizerShape, sinusoids

_lib.register_signatures({
    "astts_op_affine_act": (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32, c_int32, c_void_p]),
    "astts_op_freq_unfold": (c_int32, [c_void_p, c_int32, c_void_p] + [c_int32] * 7 + [c_void_p]),
    "astts_op_ftc_to_tfc": (c_int32, [c_void_p, c_void_p] + [c_int32] * 4 + [c_void_p]),
    "astts_op_cam_context": (c_int32, [c_void_p, c_int32, c_int64, c_void_p] + [c_int32] * 4 + [c_void_p]),
    "astts_op_cam_gate": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64] + [c_int32] * 4 + [c_void_p]),
    "astts_op_stats_pool": (c_int32, [c_void_p, c_int64, c_void_p] + [c_int32] * 3 + [c_void_p]),
    "astts_op_l2_normalize": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "astts_op_sub_time_mean": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
})

SD = Dict[str, torch.Tensor]


def _L():
    return _lib.load()


# ------------------------------------------------------------------------------------------ operator wrappers
def affine_act(x: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], relu: bool = True,
               out_dtype=torch.float16, cols: Optional[int] = None) -> torch.Tensor:
    """``act(x[:, :cols] * scale + shift)`` for a 2-D (possibly wider) buffer ``x`` [rows, ld] -> contiguous [rows, cols]."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float32, torch.float16)
    rows, c = x.shape[0], (cols if cols is not None else x.shape[1])
    y = torch.empty((rows, c), dtype=out_dtype, device=x.device)
    _lib.check(_L().astts_op_affine_act(x.data_ptr(), 1 if x.dtype == torch.float16 else 0, x.stride(0), ops._p(scale), ops._p(shift),
                                        y.data_ptr(), 1 if out_dtype == torch.float16 else 0, c, rows, c, 1 if relu else 0, _lib.stream_ptr()))
    return y


def freq_unfold(x: torch.Tensor, f_out: int, sf: int, nkf: int) -> torch.Tensor:
    """x [B, F, T, C] (fp32 / fp16) -> fp16 [B, f_out, T, nkf * C]: the frequency rows of each window side by side."""
    assert x.is_contiguous() and x.dim() == 4
    b, f, t, c = x.shape
    y = torch.empty((b, f_out, t, nkf * c), dtype=torch.float16, device=x.device)
    _lib.check(_L().astts_op_freq_unfold(x.data_ptr(), 1 if x.dtype == torch.float16 else 0, y.data_ptr(), b, f, t, c, f_out, sf, nkf,
                                         _lib.stream_ptr()))
    return y


def ftc_to_tfc(x: torch.Tensor) -> torch.Tensor:
    b, f, t, c = x.shape
    y = torch.empty((b, t, f * c), dtype=torch.float32, device=x.device)
    _lib.check(_L().astts_op_ftc_to_tfc(ops._f32(x).data_ptr(), y.data_ptr(), b, f, t, c, _lib.stream_ptr()))
    return y


def cam_context(h: torch.Tensor, seg_len: int) -> torch.Tensor:
    """h [B, T, C] -> [B, ceil(T / seg_len), C]: mean over time + segment mean."""
    assert h.is_contiguous() and h.dim() == 3
    b, t, c = h.shape
    ctx = torch.empty((b, (t + seg_len - 1) // seg_len, c), dtype=torch.float32, device=h.device)
    _lib.check(_L().astts_op_cam_context(h.data_ptr(), 1 if h.dtype == torch.float16 else 0, c, ctx.data_ptr(), b, t, c, seg_len,
                                         _lib.stream_ptr()))
    return ctx


def cam_gate(y: torch.Tensor, m: torch.Tensor, out: torch.Tensor, seg_len: int) -> None:
    """out[b, t, :] = y[b, t, :] * sigmoid(m[b, t // seg_len, :]); ``out`` a [B, T, C] view whose row stride may exceed C."""
    b, t, c = y.shape
    assert out.shape == y.shape and out.stride(2) == 1 and out.stride(0) == t * out.stride(1)
    _lib.check(_L().astts_op_cam_gate(ops._f32(y).data_ptr(), ops._f32(m).data_ptr(), out.data_ptr(), out.stride(1), b, t, c, seg_len,
                                      _lib.stream_ptr()))


def stats_pool(x: torch.Tensor) -> torch.Tensor:
    x = ops._f32(x)
    b, t, c = x.shape
    out = torch.empty((b, 2 * c), dtype=torch.float32, device=x.device)
    _lib.check(_L().astts_op_stats_pool(x.data_ptr(), c, out.data_ptr(), b, t, c, _lib.stream_ptr()))
    return out


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    x = ops._f32(x)
    y = torch.empty_like(x)
    _lib.check(_L().astts_op_l2_normalize(x.data_ptr(), y.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], eps, _lib.stream_ptr()))
    return y


def sub_time_mean_(x: torch.Tensor) -> torch.Tensor:
    """x [B, T, C] fp32 on the GPU: minus its mean over time, in place."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    _lib.check(_L().astts_op_sub_time_mean(x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], _lib.stream_ptr()))
    return x


def _bn_fold(sd: SD, p: str, eps: float):
    """eval BatchNorm as (scale, shift) per channel"""
    s = (sd[p + ".running_var"].double() + eps).rsqrt()
    if (p + ".weight") in sd:
        s = s * sd[p + ".weight"].double()
    shift = -sd[p + ".running_mean"].double() * s
    if (p + ".bias") in sd:
        shift = shift + sd[p + ".bias"].double()
    return s.float(), shift.float()


# ------------------------------------------------------------------------------------------ speech tokenizer
class SpeechTokenizerV1:
    """16 kHz prompt -> speech tokens at 50 Hz (int32 ``[1, T]``), as ``frontend._extract_speech_token`` [EXT] gets them from
    speech_tokenizer_v1.onnx."""

    learned = True

    def __init__(self, sd: SD, cfg: SpeechTokenizerShape, device, synthetic: bool = False):
        from .knn import StyleBank
        if cfg.d % cfg.heads or cfg.d // cfg.heads != 64:
            raise ValueError(f"SpeechTokenizerV1: head dimension {cfg.d // cfg.heads} (the attention kernel serves 64)")
        self.cfg, self.device, self.synthetic = cfg, torch.device(device), synthetic
        dev = self.device
        f = lambda k: sd[k].to(torch.float32)
        with torch.cuda.device(dev):
            self.conv1 = ops.PackedWeight.from_conv1d(f("encoder.conv1.weight"), f("encoder.conv1.bias"), dev)
            self.conv2 = ops.PackedWeight.from_conv1d(f("encoder.conv2.weight"), f("encoder.conv2.bias"), dev)
            self.pos = sinusoids(cfg.n_ctx, cfg.d).to(dev)
            self.blocks = []
            for i in range(cfg.layers):
                p = f"encoder.blocks.{i}."
                wqkv = torch.cat([f(p + "attn.query.weight"), f(p + "attn.key.weight"), f(p + "attn.value.weight")], 0)
                bqkv = torch.cat([f(p + "attn.query.bias"), torch.zeros(cfg.d), f(p + "attn.value.bias")], 0)
                self.blocks.append(dict(
                    ln1=(f(p + "attn_ln.weight").to(dev), f(p + "attn_ln.bias").to(dev)),
                    qkv=ops.PackedWeight(wqkv, bqkv, dev),
                    out=ops.PackedWeight(f(p + "attn.out.weight"), f(p + "attn.out.bias"), dev),
                    ln2=(f(p + "mlp_ln.weight").to(dev), f(p + "mlp_ln.bias").to(dev)),
                    fc1=ops.PackedWeight(f(p + "mlp.0.weight"), f(p + "mlp.0.bias"), dev),
                    fc2=ops.PackedWeight(f(p + "mlp.2.weight"), f(p + "mlp.2.bias"), dev)))
            self.codebook = StyleBank(f("quantizer._codebook.embed"), device=dev, metric="L2")

    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel ``[B, n_mels, T]`` on the GPU (B prompts of ONE length: no padding mask is involved, so a row does not depend on the
        batch it runs in) -> encoder frames fp32 ``[ceil(T / 2), d]`` for B = 1, ``[B, ceil(T / 2), d]`` otherwise."""
        cfg = self.cfg
        assert mel.is_cuda and mel.dim() == 3 and mel.shape[1] == cfg.n_mels
        nb, t = int(mel.shape[0]), int(mel.shape[2])
        t2 = (t - 1) // 2 + 1
        if t2 > cfg.n_ctx:
            raise ValueError(f"speech tokenizer: {t2} frames exceed the {cfg.n_ctx} positions of the encoder")
        with torch.cuda.device(self.device):
            x = mel.transpose(1, 2).contiguous()                                     # [B, T, n_mels] channels-last
            x = ops.conv1d(x, self.conv1, pad=1, act="gelu", out_dtype=torch.float16)
            pos = self.pos[:t2] if nb == 1 else self.pos[:t2].repeat(nb, 1)           # the positions ride on conv2's residual operand
            x = ops.conv1d(x, self.conv2, stride=2, pad=1, act="gelu", residual=pos).view(nb * t2, cfg.d)
            d, h = cfg.d, cfg.heads
            for b in self.blocks:
                y = ops.layernorm(x, b["ln1"][0], b["ln1"][1], 1e-5, out_dtype=torch.float16)
                qkv = ops.linear(y, b["qkv"], out_dtype=torch.float16).view(nb, t2, 3 * d)
                a = ops.attn_mha(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], h, out_dtype=torch.float16)
                x = ops.linear(a.view(nb * t2, d), b["out"], residual=x)
                y = ops.layernorm(x, b["ln2"][0], b["ln2"][1], 1e-5, out_dtype=torch.float16)
                y = ops.linear(y, b["fc1"], act="gelu", out_dtype=torch.float16)
                x = ops.linear(y, b["fc2"], residual=x)
        return x if nb == 1 else x.view(nb, t2, cfg.d)

    def quantize(self, frames: torch.Tensor) -> torch.Tensor:
        """fp32 ``[T, d]`` -> int32 ``[T]``: arg-min squared distance of the normalised frame (ties: the lower code)."""
        with torch.cuda.device(self.device):
            x = l2_normalize(frames) if self.cfg.normalize else frames
            idx, _ = self.codebook.search_device(x, 1)
            return idx[:, 0].to(torch.int32)

    def tokens_from_mel(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, n_mels, T] -> int32 [T'] (B = 1) or [B, T']"""
        x = self.encode(mel)
        q = self.quantize(x.reshape(-1, self.cfg.d))
        return q if x.dim() == 2 else q.view(x.shape[0], x.shape[1])

    def tokens_device(self, wav16k: torch.Tensor) -> torch.Tensor:
        """wav [B, n] (B prompts of one length) -> int32 [B, T'] on the GPU, no synchronisation"""
        if wav16k.shape[-1] > 30 * 16000:
            raise ValueError("do not support extract speech token for audio longer than 30s")  # upstream assert
        mel = audio.whisper_log_mel(wav16k.to(self.device), n_mels=self.cfg.n_mels)
        return self.tokens_from_mel(mel).view(mel.shape[0], -1)

    def __call__(self, wav16k: torch.Tensor) -> torch.Tensor:
        return self.tokens_device(wav16k).cpu()


# ------------------------------------------------------------------------------------------ CAM++
class _Conv2dAsTaps:
    """A 3 x 3 (or 1 x 1) Conv2d of the FCM head with its eval BatchNorm folded in, as a GEMM over frequency-unfolded rows."""

    def __init__(self, w: torch.Tensor, bn, stride_f: int, device):
        co, ci, kf, kt = (int(s) for s in w.shape)
        scale, shift = bn
        w = w.to(torch.float32) * scale[:, None, None, None]
        self.nkf, self.kt, self.sf, self.ci, self.co = kf, kt, stride_f, ci, co
        self.w = ops.PackedWeight(w.permute(0, 3, 2, 1).reshape(co, kt, kf * ci).contiguous(), shift, device)

    def __call__(self, x: torch.Tensor, act: str = "none", residual: Optional[torch.Tensor] = None, out_dtype=torch.float32) -> torch.Tensor:
        """x [B, F, T, ci] -> [B, F_out, T, co]"""
        b, f, t, _ = x.shape
        f_out = (f + 2 * ((self.nkf - 1) // 2) - self.nkf) // self.sf + 1
        u = freq_unfold(x, f_out, self.sf, self.nkf).view(b * f_out, t, self.nkf * self.ci)
        res = None if residual is None else residual.reshape(b * f_out * t, self.co)
        y = ops.conv1d(u, self.w, pad=(self.kt - 1) // 2, act=act, residual=res, out_dtype=out_dtype)
        return y.view(b, f_out, t, self.co)


class CamPlusSpeakerNet:
    """16 kHz prompt -> speaker embedding fp32 ``[1, emb]``, as ``frontend._extract_spk_embedding`` [EXT] gets it from campplus.onnx."""

    learned = True

    def __init__(self, sd: SD, cfg: CamPlusShape, device, synthetic: bool = False):
        self.cfg, self.device, self.synthetic = cfg, torch.device(device), synthetic
        dev, eps = self.device, cfg.bn_eps
        f = lambda k: sd[k].to(torch.float32)
        bn = lambda p: _bn_fold(sd, p, eps)
        with torch.cuda.device(dev):
            self.h_conv1 = _Conv2dAsTaps(f("head.conv1.weight"), bn("head.bn1"), 1, dev)
            self.h_blocks = []
            for li in (1, 2):
                for bi in (0, 1):
                    p = f"head.layer{li}.{bi}."
                    s = 2 if bi == 0 else 1
                    sc = _Conv2dAsTaps(f(p + "shortcut.0.weight"), bn(p + "shortcut.1"), s, dev) if (p + "shortcut.0.weight") in sd else None
                    self.h_blocks.append((_Conv2dAsTaps(f(p + "conv1.weight"), bn(p + "bn1"), s, dev),
                                          _Conv2dAsTaps(f(p + "conv2.weight"), bn(p + "bn2"), 1, dev), sc))
            self.h_conv2 = _Conv2dAsTaps(f("head.conv2.weight"), bn("head.bn2"), 2, dev)
            # TDNN input features: this layout is [f][c], torch's reshape of [B, C, F, T] is [c][f]
            m, fo = cfg.m_channels, cfg.feat_dim // 8
            s, sh = bn("xvector.tdnn.nonlinear.batchnorm")
            w = f("xvector.tdnn.linear.weight") * s[:, None, None]                     # [ch, c * fo + f, 5]
            w = w.view(w.shape[0], m, fo, w.shape[2]).permute(0, 2, 1, 3).reshape(w.shape[0], fo * m, w.shape[2])
            self.tdnn = ops.PackedWeight.from_conv1d(w, sh, dev)
            self.blocks = []
            ch = cfg.init_channels
            for bi, (layers, k, dil) in enumerate(cfg.blocks):
                ls = []
                for li in range(layers):
                    p = f"xvector.block{bi + 1}.tdnnd{li + 1}."
                    cin = ch + li * cfg.growth
                    s1, b1 = bn(p + "nonlinear1.batchnorm")
                    s2, b2 = bn(p + "nonlinear2.batchnorm")
                    ls.append(dict(cin=cin, s1=s1.to(dev), b1=b1.to(dev),
                                   lin1=ops.PackedWeight(f(p + "linear1.weight")[:, :, 0] * s2[:, None], b2, dev),
                                   local=ops.PackedWeight.from_conv1d(f(p + "cam_layer.linear_local.weight"), None, dev),
                                   c1=ops.PackedWeight(f(p + "cam_layer.linear1.weight")[:, :, 0], f(p + "cam_layer.linear1.bias"), dev),
                                   c2=ops.PackedWeight(f(p + "cam_layer.linear2.weight")[:, :, 0], f(p + "cam_layer.linear2.bias"), dev)))
                ch_out = ch + layers * cfg.growth
                p = f"xvector.transit{bi + 1}."
                st, bt = bn(p + "nonlinear.batchnorm")
                self.blocks.append(dict(layers=ls, k=k, dil=dil, ch_in=ch, ch_out=ch_out, ts=st.to(dev), tb=bt.to(dev),
                                        transit=ops.PackedWeight(f(p + "linear.weight")[:, :, 0], None, dev)))
                ch = ch_out // 2
            so, bo = bn("xvector.out_nonlinear.batchnorm")
            self.out_s, self.out_b = so.to(dev), bo.to(dev)
            sd_, bd_ = bn("xvector.dense.nonlinear.batchnorm")
            self.dense = ops.PackedWeight(f("xvector.dense.linear.weight")[:, :, 0] * sd_[:, None], bd_, dev)
            self.ch_final = ch

    def head(self, fbank: torch.Tensor) -> torch.Tensor:
        """fbank [B, T, F] -> TDNN input features [B, T, (F / 8) * 32]  (feature index f * 32 + c)"""
        x = fbank.transpose(1, 2).contiguous()[..., None]                    # [B, F, T, 1]
        x = self.h_conv1(x, act="relu")
        for c1, c2, sc in self.h_blocks:
            o = c1(x, act="relu", out_dtype=torch.float16)
            short = sc(x) if sc is not None else x
            o = c2(o, residual=short)                                         # bn2(conv2(.)) + shortcut
            x = affine_act(o.view(-1, o.shape[-1]), None, None, relu=True, out_dtype=torch.float32).view(o.shape)
        x = self.h_conv2(x, act="relu")
        return ftc_to_tfc(x)

    def frames(self, feats: torch.Tensor) -> torch.Tensor:
        """TDNN input features [B, T, head_out] -> frame-level features [B, T / 2, ch_final] (after out_nonlinear)"""
        cfg = self.cfg
        b, t, _ = feats.shape
        t2 = (t + 4 - 4 - 1) // 2 + 1
        first = self.blocks[0]
        buf = torch.empty((b, t2, first["ch_out"]), dtype=torch.float32, device=feats.device)
        rows = b * t2
        # TDNN (k 5, stride 2) + BN + ReLU straight into the first block's concatenation buffer
        ops.gemm(feats, self.tdnn, act="relu", t_in=t, t_out=t2, stride=2, dil=1, pad=2, out=buf.view(rows, -1)[:, :first["ch_in"]])
        for bi, blk in enumerate(self.blocks):
            flat = buf.view(rows, blk["ch_out"])
            for ly in blk["layers"]:
                cin = ly["cin"]
                a = affine_act(flat, ly["s1"], ly["b1"], relu=True, out_dtype=torch.float16, cols=cin)             # BN-ReLU of what is there so far
                h = ops.gemm(a, ly["lin1"], act="relu", out_dtype=torch.float16)                                     # 1x1 + BN-ReLU
                y = ops.conv1d(h.view(b, t2, -1), ly["local"], dil=blk["dil"], pad=(blk["k"] - 1) // 2 * blk["dil"])
                ctx = cam_context(h.view(b, t2, -1), cfg.seg_len)
                m = ops.gemm(ops.gemm(ctx.view(-1, ctx.shape[-1]), ly["c1"], act="relu"), ly["c2"])
                cam_gate(y, m.view(b, ctx.shape[1], -1), buf[:, :, cin:cin + cfg.growth], cfg.seg_len)
            a = affine_act(flat, blk["ts"], blk["tb"], relu=True, out_dtype=torch.float16)
            nxt = self.blocks[bi + 1]["ch_out"] if bi + 1 < len(self.blocks) else blk["ch_out"] // 2
            nbuf = torch.empty((b, t2, nxt), dtype=torch.float32, device=feats.device)
            ops.gemm(a, blk["transit"], out=nbuf.view(rows, nxt)[:, :blk["ch_out"] // 2], use_bias=False)
            buf = nbuf
        x = affine_act(buf.view(rows, -1), self.out_s, self.out_b, relu=True, out_dtype=torch.float32)
        return x.view(b, t2, -1)

    def embed(self, fbank: torch.Tensor) -> torch.Tensor:
        """fbank fp32 [B, T, feat_dim] on the GPU (mean over time removed) -> [B, emb]"""
        with torch.cuda.device(self.device):
            st = stats_pool(self.frames(self.head(fbank)))
            return ops.gemm(st, self.dense)

    def embed_device(self, wav16k: torch.Tensor) -> torch.Tensor:
        """wav [B, n] (one length) -> fp32 [B, emb] on the GPU, no synchronisation"""
        fb = audio.kaldi_fbank(wav16k.to(self.device), n_mels=self.cfg.feat_dim)
        return self.embed(sub_time_mean_(fb))                 # (upstream's frontend: feat - feat.mean(dim=0) in front of campplus)

    def __call__(self, wav16k: torch.Tensor) -> torch.Tensor:
        return self.embed_device(wav16k).cpu()


# ------------------------------------------------------------------------------------------ construction from a model directory
def shapes_for(cfg) -> tuple:
    """(SpeechTokenizerShape, CamPlusShape) for a SynthConfig: the published networks for the CosyVoice-300M sizes, toy ones for
    toy configs (``SynthConfig.tiny()``: 256 codes, 32-d speaker vectors)."""
    if cfg.speech_vocab == 4096 and cfg.spk_dim == 192:
        return SpeechTokenizerShape(), CamPlusShape()
    from dataclasses import replace
    return replace(SpeechTokenizerShape.tiny(), codes=cfg.speech_vocab), replace(CamPlusShape.tiny(), emb=cfg.spk_dim)



