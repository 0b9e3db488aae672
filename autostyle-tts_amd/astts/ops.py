"""Host side of the astts_op_* operators: thin ctypes wrappers over libastts.so.

torch is used for device memory and streams only; every function here enqueues hand-written HIP
kernels on the current stream and returns torch tensors that alias freshly allocated HBM.
Layout convention: fp32 activations, channels-last ``[B, T, C]`` (or ``[rows, C]``), contiguous.
"""
from __future__ import annotations

import ctypes
import math
from ctypes import c_float, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

from . import _lib

_SIGS = {
    "astts_op_pack_weight": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_gemm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                c_int32, c_int32, c_float, c_float, c_void_p]),
    "astts_op_gemm_ex": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                   c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_int32, c_float, c_float, c_void_p]),
    "astts_op_gemm_fused": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p] + [c_int32] * 11 + [c_float, c_float, c_void_p]),
        "astts_op_gemm_fused_ws": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p] + [c_int32] * 11 + [c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "astts_op_gemm_fused_workspace_bytes": (c_size_t, []),
    "astts_op_gemm_set_ring_mode": (c_int32, [c_int32]),
    "astts_op_gemm_ln": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64,
                                   c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_attn_relpos_ex": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p] + [c_int32] * 8 + [c_int64] * 3 + [c_int32] * 3 + [c_float, c_void_p]),
    "astts_prof_enable": (c_int32, [c_int32, c_int32, c_int32]),
    "astts_prof_read": (c_int32, [c_int32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64),
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)]),
    "astts_op_layernorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "astts_op_groupnorm_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "astts_op_groupnorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                     c_int32, c_int32, c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    "astts_op_elementwise": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                       c_int32, c_float, c_float, c_void_p]),
    "astts_op_embedding": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "astts_op_interp_linear": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_interp_linear_ex": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "astts_op_time_embedding": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "astts_op_attn_relpos": (c_int32, [c_void_p] * 8 + [c_int32] * 8 + [c_int64] * 3 + [c_int32] * 3 + [c_float, c_void_p]),
    "astts_op_attn_mha_ex": (c_int32, [c_void_p] * 3 + [c_int32, c_void_p, c_void_p] + [c_int32] * 7 + [c_float, c_void_p]),
    "astts_op_layernorm_ex": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "astts_op_layernorm_relu": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_float, c_float, c_void_p]),
    "astts_op_groupnorm_ex": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                        c_int32, c_int32, c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    "astts_op_attn_mha": (c_int32, [c_void_p] * 5 + [c_int32] * 6 + [c_float, c_void_p]),
    "astts_op_nsf_source_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "astts_op_nsf_source": (c_int32, [c_void_p] * 6 + [c_int32] * 4 + [c_float] * 4 + [c_void_p, c_size_t, c_void_p]),
    "astts_op_stft16": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "astts_op_istft16": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_float, c_float, c_void_p]),
    "astts_op_stft16_lens": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "astts_op_istft16_lens": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_float, c_float, c_void_p, c_void_p]),
    "astts_op_gemm_rows": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                     c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_gemm_lens": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                     c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "astts_op_ras_sample": (c_int32, [c_void_p] * 4 + [c_int32] * 5 + [c_float, c_int32, c_float, c_int32, c_int32, c_void_p]),
}


class LmConfig(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("d", "heads", "ffn", "layers", "vocab_out", "speech_vocab", "pos_center", "pos_ld",
                                       "top_k", "ras_win")] + [(n, c_float) for n in ("top_p", "ras_tau", "eps")] + \
               [("kv_f16", c_int32), ("pos_f16", c_int32), ("ln_folded", c_int32), ("eos_policy", c_int32)]


class LmGlobals(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("speech_emb", "embed_w", "embed_b", "embed_ln_g", "embed_ln_b", "after_g", "after_b",
                                        "head_w", "head_b", "embed_table")]


class LmLayer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("n1_g", "n1_b", "wqkv", "bqkv", "wo", "bo", "n2_g", "n2_b", "w1", "b1", "w2", "b2",
                                        "pos", "bias_u", "bias_v")]


_SIGS.update({
    "astts_op_ras_sample_ex": (c_int32, [c_void_p] * 4 + [c_int32] * 5 + [c_float, c_int32, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "astts_lm_create": (c_int32, [ctypes.POINTER(LmConfig), ctypes.POINTER(LmGlobals), ctypes.POINTER(LmLayer), ctypes.POINTER(c_void_p)]),
    "astts_lm_destroy": (c_int32, [c_void_p]),
    "astts_lm_workspace_bytes": (c_size_t, [c_void_p, c_int32]),
    "astts_lm_decode": (c_int32, [c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                  c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "astts_lm_decode_range": (c_int32, [c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                        c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
})


class Weight(ctypes.Structure):
    """astts_weight_t: packed fp16 weight image + fp32 bias."""
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("n", c_int32), ("cin", c_int32), ("cin_pad", c_int32), ("taps", c_int32)]


class FlowResnet(ctypes.Structure):
    _fields_ = [("c1", Weight), ("mlp", Weight), ("c2", Weight), ("res", Weight),
                ("g1_w", c_void_p), ("g1_b", c_void_p), ("g2_w", c_void_p), ("g2_b", c_void_p),
                ("c1_frag", c_void_p), ("c2_frag", c_void_p), ("res_frag", c_void_p)]


class FlowTfm(ctypes.Structure):  # astts_flow_tfm_t
    _fields_ = [("n1_w", c_void_p), ("n1_b", c_void_p), ("n3_w", c_void_p), ("n3_b", c_void_p),
                ("qkv", Weight), ("wo", Weight), ("w1", Weight), ("w2", Weight), ("qkv_frag", c_void_p),
                ("w1_frag", c_void_p), ("w2_frag", c_void_p), ("wo_frag", c_void_p)]


class FlowBlock(ctypes.Structure):
    _fields_ = [("res", FlowResnet), ("tfm", ctypes.POINTER(FlowTfm)), ("n_tfm", c_int32), ("resample", Weight),
                ("resample_kind", c_int32)]


class FlowConfig(ctypes.Structure):
    _fields_ = [("mel", c_int32), ("channels", c_int32), ("heads", c_int32), ("groups", c_int32), ("time_in", c_int32),
                ("time_dim", c_int32), ("n_down", c_int32), ("n_mid", c_int32), ("n_up", c_int32),
                ("t1", Weight), ("t2", Weight), ("fin_c", Weight), ("fin_p", Weight), ("fin_g_w", c_void_p), ("fin_g_b", c_void_p)]


FLOW_RESAMPLE_NONE, FLOW_RESAMPLE_CONV, FLOW_RESAMPLE_DOWN, FLOW_RESAMPLE_UP = range(4)

_SIGS.update({
    "astts_stream_spin": (c_int32, [c_int32, c_void_p]),
    "astts_stream_chain": (c_int32, [c_int32, c_int32, c_int32, c_void_p]),
    "astts_stream_create_cu_mask": (c_int32, [ctypes.POINTER(ctypes.c_uint32), c_int32, ctypes.POINTER(c_void_p)]),
    "astts_stream_destroy": (c_int32, [c_void_p]),
    "astts_selftest_xlane": (c_int32, [ctypes.POINTER(c_int32), c_void_p]),
    "astts_flow_create": (c_int32, [ctypes.POINTER(FlowConfig), ctypes.POINTER(FlowBlock), ctypes.POINTER(FlowBlock),
                                    ctypes.POINTER(FlowBlock), ctypes.POINTER(c_void_p)]),
    "astts_flow_destroy": (c_int32, [c_void_p]),
    "astts_flow_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32]),
    "astts_flow_solve": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                   ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_float, c_void_p, c_size_t, c_void_p]),
})
_SIGS.update({   # fused transformer-block front half of the flow estimator (csrc/ops_tfm_fused.hip)
    "astts_op_tfm_attn_fused_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "astts_op_tfm_pack_frag": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    # the flow engine's forms (L2 prefetch of the next launch's weights): called from C++, declared here for the ABI table
    "astts_op_tfm_attn_fused_pf": (c_int32, [c_void_p] * 5 + [c_int32] * 4 + [c_float, c_float, c_void_p, c_void_p, c_int32, c_void_p]),
    "astts_op_tfm_ffn_fused_pf": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_int32,
                                            c_void_p, ctypes.c_uint32, c_void_p]),
    "astts_op_conv_pack_frag": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_resnet_conv_stats_floats": (c_size_t, [c_int32, c_int32]),
    "astts_op_resnet_conv_supported": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "astts_op_resnet_conv": (c_int32, [c_void_p] * 14 + [c_int32] * 4 + [c_float, c_void_p]),
    "astts_op_resnet_conv_pf": (c_int32, [c_void_p] * 14 + [c_int32] * 4 + [c_float, c_void_p, ctypes.c_uint32, c_void_p]),
    "astts_op_conv1d_snake_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "astts_op_conv1d_snake": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_float,
                                        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_conv1d_snake_lens": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_float,
                                             c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "astts_op_tfm_ffn_fused_supported": (c_int32, [c_int32, c_int32]),
    "astts_op_tfm_ffn_fused": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "astts_op_tfm_attn_fused": (c_int32, [c_void_p] * 5 + [c_int32] * 4 + [c_float, c_float, c_void_p]),
})
_SIGS.update({   # query-embedder operators (csrc/ops_llm.hip)
    "astts_op_rmsnorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "astts_op_rope_llama": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p]),
    "astts_op_attn_causal_gqa": (c_int32, [c_void_p] * 5 + [c_int32] * 8 + [c_float, c_void_p]),
    "astts_op_swiglu": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_mean_pool": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "astts_op_attn_gqa": (c_int32, [c_void_p] * 6 + [c_int32] * 7 + [c_int64] * 6 + [c_float, c_void_p]),
    "astts_op_rope_llama_ex": (c_int32, [c_void_p] * 4 + [c_int32] * 7 + [c_void_p]),
    "astts_op_argmax_rows": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p]),
})
_lib.register_signatures(_SIGS)

ACT = {"none": 0, "relu": 1, "silu": 2, "swish": 2, "gelu": 3, "mish": 4, "elu": 5, "tanh": 6, "leaky": 7}
EL_SNAKE, EL_LEAKY, EL_ADD, EL_MUL_ROWMASK, EL_ADD_BC, EL_SCALE, EL_CFG_EULER, EL_MISH, EL_SILU, EL_CLAMP, EL_TANH, EL_ELU, EL_RELU_SCALE = range(13)


def _L():
    return _lib.load()


def _st():
    return _lib.stream_ptr()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda and t.dtype == torch.float32, (t.device, t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def _up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class PackedWeight:
    """fp16 weight image for astts_op_gemm: ``[n_pad, taps, cin_pad]`` (K contiguous, zero padded)."""

    def __init__(self, w: torch.Tensor, bias: Optional[torch.Tensor] = None, device=None):
        """``w``: fp32 ``[n, cin]`` (linear) or ``[n, taps, cin]`` (conv, cin innermost)."""
        device = device or torch.device("cuda", torch.cuda.current_device())
        if w.dim() == 2:
            w = w[:, None, :]
        assert w.dim() == 3
        self.n, self.taps, self.cin = (int(s) for s in w.shape)
        self.cin_pad = _up(self.cin, 64)
        self.n_pad = _up(self.n, 128)
        src = w.to(device=device, dtype=torch.float32).contiguous()
        self.data = torch.empty((self.n_pad, self.taps, self.cin_pad), dtype=torch.float16, device=device)
        _lib.check(_L().astts_op_pack_weight(src.data_ptr(), self.data.data_ptr(), self.n, self.taps, self.cin,
                                             self.n_pad, self.cin_pad, _st()))
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()

    @staticmethod
    def from_conv1d(weight: torch.Tensor, bias=None, device=None) -> "PackedWeight":
        """torch Conv1d weight ``[cout, cin, k]`` -> taps-major ``[cout, k, cin]``."""
        return PackedWeight(weight.permute(0, 2, 1).contiguous(), bias, device)

    @staticmethod
    def from_conv_transpose1d(weight: torch.Tensor, bias, stride: int, device=None) -> "PackedWeight":
        """torch ConvTranspose1d weight ``[cin, cout, k]`` with k == 2*stride -> phase-decomposed
        GEMM weight ``[stride*cout, 2, cin]``: output phase r of input step q uses taps
        x[q] * W[:, :, r] + x[q-1] * W[:, :, r + stride]  (see conv_transpose1d)."""
        cin, cout, k = (int(s) for s in weight.shape)
        assert k == 2 * stride, "phase decomposition implemented for kernel == 2*stride"
        w = weight.to(torch.float32)
        # gemm taps: tap 0 reads row (q - 1) [pad = 1], tap 1 reads row q
        w0 = w[:, :, stride:].permute(2, 1, 0)  # [r, cout, cin] for x[q-1]
        w1 = w[:, :, :stride].permute(2, 1, 0)  # [r, cout, cin] for x[q]
        packed = torch.stack([w0, w1], dim=2).reshape(stride * cout, 2, cin)  # n = r*cout + co
        b = None if bias is None else bias.to(torch.float32).repeat(stride)
        pw = PackedWeight(packed, b, device)
        pw.ct_stride, pw.ct_cout = stride, cout
        return pw


def _act_in(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda and t.dtype in (torch.float32, torch.float16), (t.device, t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def gemm(x: torch.Tensor, w: PackedWeight, act: str = "none", residual: Optional[torch.Tensor] = None,
         row_scale: Optional[torch.Tensor] = None, alpha: float = 1.0, slope: float = 0.1,
         t_in: Optional[int] = None, t_out: Optional[int] = None, stride: int = 1, dil: int = 1, pad: int = 0,
         out: Optional[torch.Tensor] = None, use_bias: bool = True, out_dtype=torch.float32,
         in_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x``: ``[..., cin]`` fp32 or fp16 (rows = batch*time); conv geometry via t_in/t_out/stride/dil/pad.
    ``out_dtype=torch.float16`` when the result only feeds MFMA consumers (another GEMM / attention).
    ``in_lens`` (int32 ``[batches]``, ragged batches): input steps at or beyond a row's length read as zero (astts_op_gemm_lens)."""
    x = _act_in(x)
    cin = x.shape[-1]
    assert cin == w.cin, (cin, w.cin)
    rows_in = x.numel() // cin
    if t_in is None:
        t_in = t_out = rows_in
        batches = 1
    else:
        batches = rows_in // t_in
        assert batches * t_in == rows_in
    m = batches * t_out
    if out is None:
        out = torch.empty((m, w.n), dtype=out_dtype, device=x.device)
    ldc = out.stride(0) if out.dim() == 2 else out.stride(-2)
    ldr = 0
    if residual is not None:
        residual = _f32(residual)
        ldr = residual.shape[-1]
    if in_lens is not None:
        assert in_lens.dtype == torch.int32 and in_lens.numel() == batches and in_lens.is_cuda
        _lib.check(_L().astts_op_gemm_lens(x.data_ptr(), 1 if x.dtype == torch.float16 else 0, w.data.data_ptr(),
                                           _p(w.bias) if use_bias else None, _p(residual), _p(row_scale), out.data_ptr(),
                                           1 if out.dtype == torch.float16 else 0, m, w.n, w.cin, w.cin_pad, w.taps, cin, ldc, ldr,
                                           t_in, t_out, stride, dil, pad, ACT[act], alpha, slope, in_lens.data_ptr(), _st()))
        return out
    _lib.check(_L().astts_op_gemm_ex(x.data_ptr(), 1 if x.dtype == torch.float16 else 0, w.data.data_ptr(),
                                     _p(w.bias) if use_bias else None, _p(residual), _p(row_scale), out.data_ptr(),
                                     1 if out.dtype == torch.float16 else 0, m, w.n, w.cin, w.cin_pad, w.taps, cin, ldc, ldr,
                                     t_in, t_out, stride, dil, pad, ACT[act], alpha, slope, _st()))
    return out


PROF_GEMM_TILE, PROF_GEMM_SKINNY, PROF_ATTN_FLASH, PROF_ATTN_DECODE = range(4)


def prof_enable(kind: int, on: bool = True, max_launches: int = 8192) -> None:
    _lib.check(_L().astts_prof_enable(kind, 1 if on else 0, max_launches))


def prof_read(kind: int):
    """-> (ms_sum, launches, work_sum, dropped) since the last read (synchronises)."""
    ms, work = ctypes.c_double(), ctypes.c_double()
    n, dropped = c_int64(), c_int64()
    _lib.check(_L().astts_prof_read(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work), ctypes.byref(dropped)))
    return float(ms.value), int(n.value), float(work.value), int(dropped.value)


_SPLITK_WS: dict = {}


def gemm_fused(x: torch.Tensor, w: PackedWeight, m: int, gather: Optional[torch.Tensor] = None, ln=None, ln_eps: float = 1e-5,
               act: str = "none", residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
               out2: Optional[torch.Tensor] = None, n_split: int = 0, alpha: float = 1.0, slope: float = 0.1,
               lda: Optional[int] = None) -> torch.Tensor:
    """Decode-sized GEMM (m <= 32 rows) with optional row gather, LayerNorm prologue and split output.
    ``x``: [rows, cin] fp32 (``gather`` int32 [m] selects rows); ``out`` [m, n or n_split], ``out2`` [m, n - n_split]
    (row strides taken from the tensors, so they may be views into larger buffers such as a KV cache)."""
    n1 = n_split if out2 is not None else w.n
    if out is None:
        out = torch.empty((m, n1), dtype=torch.float32, device=x.device)
    ga, be = (ln if ln is not None else (None, None))
    st = _st()
    ws = _SPLITK_WS.get((x.device.index, st))
    if ws is None:      # zeroed once per (device, stream): the kernel leaves its arrival counters at zero
        ws = _SPLITK_WS[(x.device.index, st)] = torch.zeros(int(_L().astts_op_gemm_fused_workspace_bytes()), dtype=torch.uint8,
                                                            device=x.device)
    _lib.check(_L().astts_op_gemm_fused_ws(x.data_ptr(), _p(gather), _p(ga), _p(be), ln_eps, w.data.data_ptr(), _p(w.bias),
                                           _p(residual), out.data_ptr(), _p(out2),
                                           1 if (out2 is not None and out2.dtype == torch.float16) else 0, m, w.n, n_split, w.cin,
                                           w.cin_pad, lda if lda is not None else x.stride(-2), out.stride(-2),
                                           out2.stride(-2) if out2 is not None else 0,
                                           residual.stride(-2) if residual is not None else 0, ACT[act], alpha, slope,
                                           ws.data_ptr(), ws.numel(), st))
    return out


def set_gemm_ring_mode(mode: int) -> None:
    """-1 auto (default), 0 ring kernel off, 1 / 2 / 3 / 4 / 5 force the 128x128 / 128x64 / 64x64 / 256x256 one-barrier / 256x256
    eight-phase ring tile (tests, tuning)."""
    _lib.check(_L().astts_op_gemm_set_ring_mode(int(mode)))


def linear(x: torch.Tensor, w: PackedWeight, act: str = "none", residual=None, alpha: float = 1.0,
           out_dtype=torch.float32) -> torch.Tensor:
    y = gemm(x.reshape(-1, x.shape[-1]), w, act=act, residual=None if residual is None else residual.reshape(-1, w.n),
             alpha=alpha, out_dtype=out_dtype)
    return y.view(*x.shape[:-1], w.n)


def linear_ln(x: torch.Tensor, w: PackedWeight, residual: torch.Tensor, ln, eps: float = 1e-5):
    """``out = x @ w^T + bias + residual`` (fp32) and ``LayerNorm(out) * gamma + beta`` (fp16) from one launch
    (astts_op_gemm_ln: x fp16 ``[..., cin]``, w.n == 256).  -> (out, ln_out)"""
    assert x.dtype == torch.float16 and x.is_contiguous() and w.n == 256 and w.taps == 1 and w.cin == w.cin_pad
    rows = x.numel() // x.shape[-1]
    res = _f32(residual).reshape(rows, w.n)
    out = torch.empty((rows, w.n), dtype=torch.float32, device=x.device)
    ln_out = torch.empty((rows, w.n), dtype=torch.float16, device=x.device)
    _lib.check(_L().astts_op_gemm_ln(x.data_ptr(), w.data.data_ptr(), _p(w.bias), res.data_ptr(), out.data_ptr(), ln[0].data_ptr(),
                                     ln[1].data_ptr(), eps, ln_out.data_ptr(), rows, w.n, w.cin, w.cin_pad, x.shape[-1], w.n, w.n, w.n, _st()))
    return out.view(*x.shape[:-1], w.n), ln_out.view(*x.shape[:-1], w.n)


def gemm_rows(x: torch.Tensor, w: PackedWeight, act: str = "none", residual: Optional[torch.Tensor] = None, out_dtype=torch.float32,
              out: Optional[torch.Tensor] = None, n: Optional[int] = None, row0: int = 0,
              out2: Optional[torch.Tensor] = None, n_split: int = 0) -> torch.Tensor:
    """``act(x @ w[row0 : row0 + n]^T + bias) + residual`` for a few hundred rows (astts_op_gemm_rows: the wide decode engine's
    projection kernel -- one memory round trip per workgroup; a row's sums do not depend on the other rows).  ``x`` fp32 or fp16
    ``[m, k]`` with k == w.cin == w.cin_pad; ``out2``: columns >= n_split go there (e.g. a KV-cache row, fp16)."""
    x = _act_in(x)
    assert x.dim() == 2 and w.taps == 1 and w.cin == w.cin_pad == x.shape[1] and x.stride(1) == 1
    m, k = x.shape
    n = w.n - row0 if n is None else n
    if out is None:
        out = torch.empty((m, n_split if out2 is not None else n), dtype=out_dtype, device=x.device)
    if residual is not None:
        residual = _f32(residual)
    bias = None if w.bias is None else w.bias[row0:]
    _lib.check(_L().astts_op_gemm_rows(x.data_ptr(), 1 if x.dtype == torch.float16 else 0, w.data.data_ptr() + 2 * row0 * w.cin_pad,
                                       _p(bias), _p(residual), out.data_ptr(), 1 if out.dtype == torch.float16 else 0, _p(out2),
                                       1 if (out2 is not None and out2.dtype == torch.float16) else 0, m, n, n_split, k, x.stride(0), out.stride(0),
                                       out2.stride(0) if out2 is not None else 0, residual.stride(0) if residual is not None else 0, ACT[act], _st()))
    return out


def conv1d(x: torch.Tensor, w: PackedWeight, stride: int = 1, dil: int = 1, pad: int = 0, act: str = "none",
           residual=None, alpha: float = 1.0, slope: float = 0.1, out_dtype=torch.float32, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x``: ``[B, T, Cin]`` -> ``[B, T_out, Cout]`` (nn.Conv1d semantics on the time axis).  ``lens`` (int32 ``[B]``): each row convolves
    as a sequence of its own length (zero padding behind it); outputs behind a row's own output length are unspecified."""
    b, t, _ = x.shape
    k = w.taps
    t_out = (t + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y = gemm(x, w, act=act, residual=residual, alpha=alpha, slope=slope, t_in=t, t_out=t_out, stride=stride, dil=dil, pad=pad,
             out_dtype=out_dtype, in_lens=lens)
    return y.view(b, t_out, w.n)


def conv_transpose1d(x: torch.Tensor, w: PackedWeight, padding: int, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.ConvTranspose1d(kernel = 2*stride, stride, padding) on ``[B, T, Cin]`` -> ``[B, stride*T, Cout]``
    (for padding == stride//2).  Phase decomposition: one GEMM over T+1 steps with two taps producing
    ``stride`` output phases per step, then a shifted view."""
    b, t, _ = x.shape
    s, cout = w.ct_stride, w.ct_cout
    y = gemm(x, w, t_in=t, t_out=t + 1, stride=1, dil=1, pad=1, in_lens=lens)  # [B*(T+1), s*cout]; ``lens``: input steps of each row
    y = y.view(b, (t + 1) * s, cout)
    t_full = (t - 1) * s - 2 * padding + 2 * s
    return y[:, padding:padding + t_full, :].contiguous()


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out_dtype=torch.float32,
              relu_scale: float = 0.0) -> torch.Tensor:
    """``relu_scale`` > 0: ``relu_scale * max(LayerNorm(x), 0)`` in the same launch."""
    x = _f32(x)
    c = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_L().astts_op_layernorm_relu(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                            1 if out_dtype == torch.float16 else 0, x.numel() // c, c, c, c, eps, relu_scale, _st()))
    return y


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float = 1e-5,
              lens: Optional[torch.Tensor] = None, mish: bool = False, add_bc: Optional[torch.Tensor] = None,
              out_dtype=torch.float32) -> torch.Tensor:
    x = _f32(x)
    b, t, c = x.shape
    need = int(_L().astts_op_groupnorm_workspace_bytes(b, t, groups))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_L().astts_op_groupnorm_ex(x.data_ptr(), _p(lens), gamma.data_ptr(), beta.data_ptr(), _p(add_bc),
                                          y.data_ptr(), 1 if out_dtype == torch.float16 else 0, b, t, c, groups, eps,
                                          1 if mish else 0, ws.data_ptr(), need, _st()))
    return y


def elementwise(op: int, x: torch.Tensor, z=None, p0=None, lens=None, s: float = 0.0, s2: float = 0.0,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _f32(x)
    c = x.shape[-1]
    t = x.shape[-2] if x.dim() >= 2 else 1
    y = out if out is not None else torch.empty_like(x)
    _lib.check(_L().astts_op_elementwise(op, x.data_ptr(), _p(z), _p(p0), _p(lens), y.data_ptr(), x.numel(), t, c,
                                         s, s2, _st()))
    return y


def embedding(table: torch.Tensor, ids: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    ids = ids.to(torch.int32).contiguous()
    c = table.shape[1]
    y = torch.empty((*ids.shape, c), dtype=torch.float32, device=table.device)
    _lib.check(_L().astts_op_embedding(table.data_ptr(), ids.data_ptr(), y.data_ptr(), ids.numel(), c, c,
                                       table.shape[0], scale, _st()))
    return y


def interp_linear(x: torch.Tensor, t_out: int, in_lens: Optional[torch.Tensor] = None,
                  out_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.interpolate(mode="linear") along T; with in_lens / out_lens each row is resampled from its own valid length
    to its own output length (ragged batch), zeros beyond."""
    x = _f32(x)
    b, t, c = x.shape
    y = torch.empty((b, t_out, c), dtype=torch.float32, device=x.device)
    _lib.check(_L().astts_op_interp_linear_ex(x.data_ptr(), y.data_ptr(), b, t, t_out, c, _p(in_lens), _p(out_lens), _st()))
    return y


def time_embedding(t: torch.Tensor, dim: int, scale: float = 1000.0) -> torch.Tensor:
    t = _f32(t)
    y = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    _lib.check(_L().astts_op_time_embedding(t.data_ptr(), y.data_ptr(), t.numel(), dim, scale, _st()))
    return y


def attn_relpos(q, k, v, pos, bias_u, bias_v, heads: int, lens=None, q_pos0: int = 0, pos_center: int = 0,
                causal: bool = False, time_major: bool = False, out: Optional[torch.Tensor] = None,
                key_start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q: [B, Tq, *] (or [Tq, B, *] when time_major) strided view, k/v: [B, Tk, *] / [Tk, B, *];
    head h occupies columns h*64..h*64+63 of each view."""
    if time_major:
        tq, b, tk = q.shape[0], q.shape[1], k.shape[0]
        ldq, q_bs, ldk, k_bs = q.stride(0), q.stride(1), k.stride(0), k.stride(1)
        if out is None:
            out = torch.empty((tq, b, heads * 64), dtype=torch.float32, device=q.device)
        ldo, o_bs = out.stride(0), out.stride(1)
    else:
        b, tq, tk = q.shape[0], q.shape[1], k.shape[1]
        ldq, q_bs, ldk, k_bs = q.stride(1), q.stride(0), k.stride(1), k.stride(0)
        if out is None:
            out = torch.empty((b, tq, heads * 64), dtype=torch.float32, device=q.device)
        ldo, o_bs = out.stride(1), out.stride(0)
    assert v.stride() == k.stride() and k.dtype == v.dtype
    _lib.check(_L().astts_op_attn_relpos_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), 1 if k.dtype == torch.float16 else 0,
                                            pos.data_ptr(), 1 if pos.dtype == torch.float16 else 0, bias_u.data_ptr(),
                                            bias_v.data_ptr(), _p(lens), _p(key_start), out.data_ptr(), b, heads, tq, tk, ldq, ldk, ldo,
                                            pos.stride(0), q_bs, k_bs, o_bs, q_pos0, pos_center, 1 if causal else 0,
                                            1.0 / math.sqrt(64.0), _st()))
    return out


def attn_mha(q, k, v, heads: int, lens=None, out_dtype=torch.float32) -> torch.Tensor:
    """q/k/v: [B, T, *] strided views (fp32 or fp16, e.g. thirds of a fused qkv buffer)."""
    b, t = q.shape[0], q.shape[1]
    out = torch.empty((b, t, heads * 64), dtype=out_dtype, device=q.device)
    assert q.stride(0) == t * q.stride(1) and k.stride(0) == t * k.stride(1) and q.dtype == k.dtype == v.dtype
    _lib.check(_L().astts_op_attn_mha_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), 1 if q.dtype == torch.float16 else 0,
                                         _p(lens), out.data_ptr(), 1 if out_dtype == torch.float16 else 0, b, heads, t,
                                         q.stride(1), k.stride(1), heads * 64, 1.0 / math.sqrt(64.0), _st()))
    return out


def tfm_attn_fused_supported(c: int, heads: int, t: int) -> bool:
    return bool(_L().astts_op_tfm_attn_fused_supported(c, heads, t))


def conv_pack_frag(w: PackedWeight) -> torch.Tensor:
    """Conv1d weight image ``[n, taps, cin]`` (``PackedWeight.from_conv1d``) in per-tap MFMA fragment order for ``conv1d_snake``."""
    assert w.cin == w.cin_pad and w.cin % 16 == 0 and w.n % 32 == 0
    out = torch.empty((w.taps, w.n, w.cin), dtype=torch.float16, device=w.data.device)
    _lib.check(_L().astts_op_conv_pack_frag(w.data.data_ptr(), out.data_ptr(), w.n, w.taps, w.cin, _st()))
    return out


def resnet_conv_supported(cin: int, cout: int, groups: int, taps: int) -> bool:
    return bool(_L().astts_op_resnet_conv_supported(cin, cout, groups, taps))


def resnet_conv(x: torch.Tensor, w: PackedWeight, w_frag: torch.Tensor, lens=None, in_gn=None, in_add=None, res_gn=None,
                want_stats: bool = False, eps: float = 1e-5):
    """One convolution of a ResnetBlock1D with the neighbouring GroupNorm + Mish folded in (csrc/ops_resnet_conv.hip).
    ``in_gn`` = (stats, gamma, beta): the input is GroupNorm -> Mish (+ ``in_add`` [B, C]) -> mask of ``x`` (applied while staging);
    ``res_gn`` = (h, stats, gamma, beta): ``mask * mish(GroupNorm(h))`` is added to the output.  Returns ``out`` or ``(out, stats)``."""
    x = _f32(x)
    b, t, c = x.shape
    assert w.cin == c and w.n == 256 and w_frag.shape == (w.taps, w.n, c)
    out = torch.empty((b, t, w.n), dtype=torch.float32, device=x.device)
    stats = torch.empty(int(_L().astts_op_resnet_conv_stats_floats(b, t)), dtype=torch.float32, device=x.device) if want_stats else None
    i_s, i_g, i_b = in_gn if in_gn is not None else (None, None, None)
    r_h, r_s, r_g, r_b = res_gn if res_gn is not None else (None, None, None, None)
    _lib.check(_L().astts_op_resnet_conv(x.data_ptr(), w_frag.data_ptr(), _p(w.bias), out.data_ptr(), _p(i_s), _p(i_g), _p(i_b), _p(in_add),
                                         _p(r_h), _p(r_s), _p(r_g), _p(r_b), _p(stats), _p(lens), b, t, c, w.taps, eps, _st()))
    return (out, stats) if want_stats else out


def conv1d_snake_supported(c: int, taps: int, dil: int) -> bool:
    return bool(_L().astts_op_conv1d_snake_supported(c, taps, dil))


def conv1d_snake(x: torch.Tensor, w: PackedWeight, w_frag: torch.Tensor, dil: int = 1, alpha: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None, out_dtype=torch.float32, want_y: bool = True,
                 acc: Optional[torch.Tensor] = None, acc_scale: float = 1.0, acc_add: bool = False, lens: Optional[torch.Tensor] = None):
    """``conv1d_same(snake_alpha(x)) + bias + residual`` on channels-last ``[B, L, C]`` (C -> C, LDS-staged kernel).  Returns ``y``
    (``out_dtype``) unless ``want_y`` is False; ``acc`` (fp32, same shape) receives ``(acc if acc_add else 0) + acc_scale * y``."""
    assert x.dtype in (torch.float32, torch.float16) and x.is_contiguous() and x.dim() == 3
    b, l, c = x.shape
    assert w.n == c and w.cin == c and w_frag.shape == (w.taps, c, c)
    y = torch.empty((b, l, c), dtype=out_dtype, device=x.device) if want_y else None
    if residual is not None:
        residual = _f32(residual)
        assert residual.shape == x.shape
    if acc is not None:
        assert acc.dtype == torch.float32 and acc.is_contiguous() and acc.shape == x.shape
    _lib.check(_L().astts_op_conv1d_snake_lens(x.data_ptr(), 1 if x.dtype == torch.float16 else 0, _p(alpha), w_frag.data_ptr(), _p(w.bias),
                                               _p(residual), _p(y), 1 if out_dtype == torch.float16 else 0, _p(acc), acc_scale,
                                               1 if acc_add else 0, b, l, c, w.taps, dil, _p(lens), _st()))
    return y


def tfm_pack_frag(w: PackedWeight) -> torch.Tensor:
    """``w.data`` (row-major fp16 [n, 1, cin]) re-ordered into MFMA fragment order: the weight image of the fused
    transformer-block kernels (``tfm_attn_fused``, ``tfm_ffn_fused``)."""
    assert w.taps == 1 and w.cin == w.cin_pad and w.cin % 16 == 0 and w.n % 32 == 0
    out = torch.empty((w.n, w.cin), dtype=torch.float16, device=w.data.device)
    _lib.check(_L().astts_op_tfm_pack_frag(w.data.data_ptr(), out.data_ptr(), w.n, w.cin, _st()))
    return out


def tfm_ffn_fused_supported(c: int, hidden: int) -> bool:
    return bool(_L().astts_op_tfm_ffn_fused_supported(c, hidden))


def tfm_ffn_fused(x: torch.Tensor, w1: PackedWeight, w1_frag: torch.Tensor, w2: PackedWeight, w2_frag: torch.Tensor, eps: float = 1e-5,
                  attn: Optional[torch.Tensor] = None, wo: Optional[PackedWeight] = None, wo_frag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x' + W2 gelu(W1 LayerNorm(x') + b1) + b2`` in one launch (LayerNorm affine folded into ``w1``): x fp32 ``[..., 256]``.
    With ``attn`` (fp16 ``[..., k0]``), ``wo`` and ``wo_frag = tfm_pack_frag(wo)``: ``x' = x + attn Wo^T + bo`` (the attention's
    output projection and residual, in the same launch); otherwise ``x' = x``."""
    x = _f32(x)
    c, hidden = x.shape[-1], w1.n
    assert w1.cin == c and w2.cin == hidden and w2.n == c and w1_frag.shape == (hidden, c) and w2_frag.shape == (c, hidden)
    out = torch.empty_like(x)
    m = x.numel() // c
    k0 = 0
    if attn is not None:
        k0 = attn.shape[-1]
        assert attn.dtype == torch.float16 and attn.is_contiguous() and attn.numel() // k0 == m
        assert wo is not None and wo.n == c and wo.cin == k0 and wo_frag.shape == (c, k0)
    _lib.check(_L().astts_op_tfm_ffn_fused(x.data_ptr(), w1_frag.data_ptr(), _p(w1.bias), w2_frag.data_ptr(), _p(w2.bias), out.data_ptr(),
                                           m, c, hidden, eps, _p(attn), _p(wo_frag), _p(wo.bias) if wo is not None else None, k0, _st()))
    return out


def tfm_attn_fused(x: torch.Tensor, wqkv: PackedWeight, wqkv_frag: torch.Tensor, heads: int, lens=None, eps: float = 1e-5) -> torch.Tensor:
    """LayerNorm (no affine: folded into ``wqkv``) + q|k|v projection + masked MHA in one launch:
    x fp32 ``[B, T, 256]`` -> fp16 ``[B, T, heads*64]``.  ``wqkv_frag = tfm_pack_frag(wqkv)``.  Caller checks
    ``tfm_attn_fused_supported`` first."""
    x = _f32(x)
    b, t, c = x.shape
    assert wqkv.cin == c == wqkv.cin_pad and wqkv.n == 3 * heads * 64 and wqkv_frag.shape == (wqkv.n, c)
    out = torch.empty((b, t, heads * 64), dtype=torch.float16, device=x.device)
    _lib.check(_L().astts_op_tfm_attn_fused(x.data_ptr(), wqkv_frag.data_ptr(), _p(wqkv.bias), _p(lens), out.data_ptr(), b, heads, t, c, eps,
                                            1.0 / math.sqrt(64.0), _st()))
    return out


def nsf_source(f0, phase0, noise, lin_w, lin_b, upsample: int, sample_rate: float, sine_amp: float, noise_std: float,
               voiced_threshold: float) -> torch.Tensor:
    b, tm = f0.shape
    nh = phase0.shape[1]
    need = int(_L().astts_op_nsf_source_workspace_bytes(b, tm))
    ws = torch.empty(need, dtype=torch.uint8, device=f0.device)
    out = torch.empty((b, tm * upsample), dtype=torch.float32, device=f0.device)
    _lib.check(_L().astts_op_nsf_source(_f32(f0).data_ptr(), _f32(phase0).data_ptr(), _f32(noise).data_ptr(),
                                        _f32(lin_w).data_ptr(), _f32(lin_b).data_ptr(), out.data_ptr(), b, tm, upsample,
                                        nh, sample_rate, sine_amp, noise_std, voiced_threshold, ws.data_ptr(), need, _st()))
    return out


def stft16(x: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``lens`` (int32 ``[B]``, samples of each row, multiples of 4): every row is transformed as a signal of its own length."""
    x = _f32(x)
    b, n = x.shape
    y = torch.empty((b, n // 4 + 1, 18), dtype=torch.float32, device=x.device)
    _lib.check(_L().astts_op_stft16_lens(x.data_ptr(), y.data_ptr(), b, n, _p(lens), _st()))
    return y


def istft16(y: torch.Tensor, mag_clip: float = 100.0, audio_limit: float = 0.99, frame_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``frame_lens`` (int32 ``[B]``): frames of each row; a row's overlap-add sees its own frames only."""
    y = _f32(y)
    b, f, _ = y.shape
    wav = torch.empty((b, 4 * (f - 1)), dtype=torch.float32, device=y.device)
    _lib.check(_L().astts_op_istft16_lens(y.data_ptr(), wav.data_ptr(), b, f, mag_clip, audio_limit, _p(frame_lens), _st()))
    return wav


def ras_sample(logits, history, hist_len: int, uniforms, top_k: int, top_p: float, win_size: int, tau_r: float,
               eos_id: int, ignore_eos: bool, out: Optional[torch.Tensor] = None, eos_policy: str = "mask") -> torch.Tensor:
    """``ignore_eos``: EOS may not be produced at this step; ``eos_policy`` says how: "mask" removes the EOS logit before the softmax,
    "reject" is upstream's re-draw until the token is not EOS (SynthConfig.eos_policy)."""
    if eos_policy not in ("mask", "reject"):
        raise ValueError(f"eos_policy {eos_policy!r}: expected 'mask' or 'reject'")
    logits = _f32(logits)
    b, v = logits.shape
    if out is None:
        out = torch.empty((b,), dtype=torch.int32, device=logits.device)
    _lib.check(_L().astts_op_ras_sample(logits.data_ptr(), _p(history), _f32(uniforms).data_ptr(), out.data_ptr(), b, v,
                                        hist_len, history.stride(0) if history is not None else 0, top_k, top_p,
                                        win_size, tau_r, eos_id, (1 if ignore_eos else 0) | (2 if eos_policy == "reject" else 0), _st()))
    return out


def selftest_xlane() -> int:
    """Number of words in which csrc/xlane.h's lane exchanges differ from __shfl_xor (0 = the decode-step reductions are intact)."""
    out = c_int32(-1)
    _lib.check(_L().astts_selftest_xlane(ctypes.byref(out), _st()))
    return int(out.value)


def cu_masked_stream(cus, device=None) -> "torch.cuda.ExternalStream":
    """A torch stream whose kernels run only on the CUs whose mask bits are listed in ``cus`` (iterable of bit indices, or an int n =
    the first n).  On gfx950 bit b enables CU b // 8 of XCC b % 8 (scripts/micro/cu_census.hip): a mask takes CUs away inside every
    XCC, never whole XCCs (an XCC whose bits are all clear keeps ALL its CUs).  The HIP stream lives as long as the process (nothing
    destroys it: a handful per experiment, scripts/cu_mask_probe*.py)."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    ids = list(range(cus)) if isinstance(cus, int) else [int(c) for c in cus]
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    if not ids or min(ids) < 0 or max(ids) >= n_cu:
        raise ValueError(f"cu_masked_stream: need a non-empty list of mask bits in 0..{n_cu - 1}, got {ids[:8]}{'...' if len(ids) > 8 else ''}")
    words = (max(ids) // 32) + 1
    arr = (ctypes.c_uint32 * words)()
    for c in ids:
        arr[c // 32] |= 1 << (c % 32)
    out = c_void_p()
    with torch.cuda.device(dev):
        _lib.check(_L().astts_stream_create_cu_mask(arr, words, ctypes.byref(out)))
    return torch.cuda.ExternalStream(out.value, device=dev)


def concurrent_streams(n: int, priority: int = 0, candidates: int = 16, device=None, protect: int = 2) -> list:
    """``n`` torch streams that pairwise overlap on the device.  HIP maps its streams onto a handful of hardware queues
    (by creation order, opaque to the caller) and two streams on one queue serialise; each candidate is probed against the
    streams already chosen with a 300 us busy-wait kernel on both (concurrent: ~300 us wall, shared queue: ~600)."""
    import time

    dev = device or torch.device("cuda", torch.cuda.current_device())
    lib = _L()

    def overlaps(a, b) -> bool:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _lib.check(lib.astts_stream_spin(300, int(a.cuda_stream)))
        _lib.check(lib.astts_stream_spin(300, int(b.cuda_stream)))
        a.synchronize()
        b.synchronize()
        return (time.perf_counter() - t0) < 480e-6

    chosen = []
    with torch.cuda.device(dev):
        pool = [torch.cuda.Stream(device=dev, priority=priority) for _ in range(max(candidates, n))]
        _lib.check(lib.astts_stream_spin(10, int(pool[0].cuda_stream)))      # module load / first-launch cost out of the way
        for s in pool:
            if all(overlaps(s, c) for c in chosen):
                chosen.append(s)
                if len(chosen) == n:
                    return chosen
    # fewer independent queues than asked for: fill up with streams that still overlap with the first `protect` chosen
    # ones (so only the later streams double up on a queue), then with anything
    for strict in (True, False):
        for s in pool:
            if len(chosen) == n:
                return chosen
            if s not in chosen and (not strict or all(overlaps(s, c) for c in chosen[:protect])):
                chosen.append(s)
    return chosen


def stream_pipe_classes(candidates: int = 12, priority: int = 0, device=None, verbose: bool = False) -> list:
    """Torch streams grouped by the command-processor pipe their hardware queue sits on: ``[[s, ...], [s, ...], ...]``.

    Two streams on different hardware queues overlap LONG kernels (what ``concurrent_streams`` probes), but queues that share a
    pipe take turns at every kernel boundary: two LAUNCH CHAINS on such a pair run 2.4x slower EACH (11 us per launch instead of
    4.6; MI355X / ROCm 7.2: four pipes, streams i and i + 4 of the pool collide).  A decode chain next to the render chain on
    one pipe is what made three-chain pipelines 152-220 ms per batch.  Probe: two chains of 300 dependent 3-us kernels, one per
    stream, enqueued from two host threads; a pair is in one class when it takes > 1.6x the slower stream's own time."""
    import threading
    import time

    dev = device or torch.device("cuda", torch.cuda.current_device())
    lib = _L()
    count, us, blocks = 300, 3, 64

    def run(group) -> float:
        torch.cuda.synchronize(dev)

        def one(st):
            _lib.check(lib.astts_stream_chain(count, us, blocks, int(st.cuda_stream)))
            st.synchronize()
        th = [threading.Thread(target=one, args=(st,)) for st in group]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return (time.perf_counter() - t0) / count

    with torch.cuda.device(dev):
        pool = [torch.cuda.Stream(device=dev, priority=priority) for _ in range(candidates)]
        alone = []
        for st in pool:
            run([st])                       # first use of a stream creates its queue: out of the way
            alone.append(min(run([st]), run([st])))
        classes, reps = [], []
        for i, st in enumerate(pool):
            home = None
            for ci, r in enumerate(reps):
                if run([pool[r], st]) > 1.6 * max(alone[r], alone[i]):
                    home = ci
                    break
            if home is None:
                classes.append([st])
                reps.append(i)
            else:
                classes[home].append(st)
    if verbose:
        print(f"stream_pipe_classes: {len(pool)} streams on {len(classes)} pipes ({[len(c) for c in classes]} per pipe); "
              f"a chain alone {min(alone) * 1e6:.1f} us per launch", flush=True)
    return classes


_PIPE_CLASSES: dict = {}
_BESIDE: dict = {}


def stream_beside(cur: Optional["torch.cuda.Stream"] = None, device=None, verbose: bool = False) -> "torch.cuda.Stream":
    """A stream whose launch chains run BESIDE those of ``cur`` (default: the current stream): another hardware queue on another
    command-processor pipe, found by measurement AGAINST ``cur`` itself.  Picking "the second pipe class" is not enough -- which queue
    the caller's own stream sits on is decided by the order streams were created in the process (in a long-lived process the render
    stream and a decode stream picked that way shared a queue: the first streamed chunk queued behind every decode range,
    BENCH_r05: 119 / 159 ms where a fresh process measured 86 / 87).  Cached per (device, stream handle): a stream keeps its queue."""
    import threading
    import time

    dev = device or torch.device("cuda", torch.cuda.current_device())
    cur = cur if cur is not None else torch.cuda.current_stream(dev)
    key = (dev.index, int(cur.cuda_stream))
    if key in _BESIDE:
        return _BESIDE[key]
    if dev.index not in _PIPE_CLASSES:
        _PIPE_CLASSES[dev.index] = stream_pipe_classes(device=dev)
    classes = _PIPE_CLASSES[dev.index]
    lib = _L()
    count, us, blocks = 200, 3, 64

    def run(group) -> float:
        torch.cuda.synchronize(dev)

        def one(st):
            _lib.check(lib.astts_stream_chain(count, us, blocks, int(st.cuda_stream)))
            st.synchronize()
        th = [threading.Thread(target=one, args=(st,)) for st in group]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return (time.perf_counter() - t0) / count

    with torch.cuda.device(dev):
        run([cur])
        alone = min(run([cur]), run([cur]))
        best, best_t = None, None
        for cl in classes:
            t = min(run([cur, cl[0]]), run([cur, cl[0]]))
            if verbose:
                print(f"stream_beside: chain beside class of {len(cl)}: {t * 1e6:.1f} us per launch (alone {alone * 1e6:.1f})", flush=True)
            if best_t is None or t < best_t:
                best, best_t = cl[0], t
            if t < 1.3 * alone:
                break
    _BESIDE[key] = best
    return best


# ---------------------------------------------------------------------------------------------- query embedder (csrc/ops_llm.hip)
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out_dtype=torch.float16) -> torch.Tensor:
    x = _f32(x)
    c = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_L().astts_op_rmsnorm(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1 if out_dtype == torch.float16 else 0,
                                     x.numel() // c, c, c, c, eps, _st()))
    return y


def rope_llama_(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, head_dim: int, pos0: int = 0) -> torch.Tensor:
    """In place on the first ``heads * head_dim`` columns of ``x`` fp16 ``[B, T, ld]`` (a strided view into q|k|v is fine)."""
    assert x.dtype == torch.float16 and x.dim() == 3 and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1)
    b, t = x.shape[0], x.shape[1]
    assert cos.shape[0] >= pos0 + t and cos.shape[1] == head_dim // 2 and cos.is_contiguous() and sin.is_contiguous()
    _lib.check(_L().astts_op_rope_llama(x.data_ptr(), cos.data_ptr(), sin.data_ptr(), b, t, heads, x.stride(1), head_dim, pos0, _st()))
    return x


def attn_causal_gqa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, kv_heads: int, head_dim: int,
                    lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q ``[B, T, heads*hd]``, k / v ``[B, T, kv_heads*hd]`` fp16 strided views -> fp16 ``[B, T, heads*hd]``."""
    b, t = q.shape[0], q.shape[1]
    assert q.dtype == k.dtype == v.dtype == torch.float16 and q.stride(0) == t * q.stride(1) and k.stride(0) == t * k.stride(1)
    assert k.stride(1) == v.stride(1)
    out = torch.empty((b, t, heads * head_dim), dtype=torch.float16, device=q.device)
    _lib.check(_L().astts_op_attn_causal_gqa(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(lens), out.data_ptr(), b, t, heads, kv_heads,
                                             head_dim, q.stride(1), k.stride(1), heads * head_dim, 1.0 / math.sqrt(head_dim), _st()))
    return out


def attn_gqa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, kv_heads: int, head_dim: int, lens: Optional[torch.Tensor] = None,
             key_start: Optional[torch.Tensor] = None, pos0: int = 0, time_major: bool = False) -> torch.Tensor:
    """Causal grouped-query attention on the matrix cores (astts_op_attn_gqa).  q ``[B, Tq, heads*hd]``, k / v ``[B, Tk, kv_heads*hd]``
    fp16 strided views (``time_major``: ``[T, B, ...]``, e.g. the first rows of a time-major KV cache); query i has key index
    ``pos0 + i``; ``lens`` masks keys at or beyond (right padding), ``key_start`` keys before (left padding).  -> fp16, laid out as q."""
    assert q.dtype == k.dtype == v.dtype == torch.float16 and q.dim() == k.dim() == v.dim() == 3
    assert q.stride(2) == k.stride(2) == v.stride(2) == 1 and k.stride() == v.stride()
    bd, td = (1, 0) if time_major else (0, 1)
    b, tq, tk = q.shape[bd], q.shape[td], k.shape[td]
    out = torch.empty(q.shape[:2] + (heads * head_dim,), dtype=torch.float16, device=q.device)
    _lib.check(_L().astts_op_attn_gqa(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(lens), _p(key_start), out.data_ptr(), b, tq, tk, pos0, heads,
                                      kv_heads, head_dim, q.stride(bd), q.stride(td), k.stride(bd), k.stride(td), out.stride(bd), out.stride(td),
                                      1.0 / math.sqrt(head_dim), _st()))
    return out


def rope_llama_ex_(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, head_dim: int, pos0: int = 0,
                   shift: Optional[torch.Tensor] = None, time_major: bool = False) -> torch.Tensor:
    """rope_llama_ with the row order stated (``time_major``: x is ``[T, B, ld]``) and a per-row position shift (int32 ``[B]``: the time
    step of row b's first real token -- left-padded prompts keep the positions transformers gives them).  In place."""
    assert x.dtype == torch.float16 and x.dim() == 3 and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1)
    t, b = (x.shape[0], x.shape[1]) if time_major else (x.shape[1], x.shape[0])
    assert cos.shape[0] >= pos0 + t and cos.shape[1] == head_dim // 2 and cos.is_contiguous() and sin.is_contiguous()
    _lib.check(_L().astts_op_rope_llama_ex(x.data_ptr(), cos.data_ptr(), sin.data_ptr(), _p(shift), b, t, 1 if time_major else 0, heads,
                                           x.stride(1), head_dim, pos0, _st()))
    return x


def argmax_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 ``[rows, n]`` -> int32 ``[rows]`` (ties: the lowest index), on the device: the greedy step's token."""
    x = _f32(x)
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty((x.shape[0],), dtype=torch.int32, device=x.device)
    _lib.check(_L().astts_op_argmax_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0), _st()))
    return out


def swiglu(gate_up: torch.Tensor) -> torch.Tensor:
    """``[..., 2f]`` fp16 (gate | up) -> ``[..., f]`` fp16 = silu(gate) * up."""
    assert gate_up.dtype == torch.float16 and gate_up.is_contiguous()
    f = gate_up.shape[-1] // 2
    out = torch.empty((*gate_up.shape[:-1], f), dtype=torch.float16, device=gate_up.device)
    _lib.check(_L().astts_op_swiglu(gate_up.data_ptr(), out.data_ptr(), gate_up.numel() // (2 * f), f, 2 * f, f, _st()))
    return out


def mean_pool(x: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _f32(x)
    b, t, c = x.shape
    out = torch.empty((b, c), dtype=torch.float32, device=x.device)
    _lib.check(_L().astts_op_mean_pool(x.data_ptr(), _p(lens), out.data_ptr(), b, t, c, _st()))
    return out
