"""GPU numerics tests of the astts_op_* HIP operators against plain PyTorch fp32 (CPU) references of
the same op.  Tolerances: fp16-operand / fp32-accumulate contractions are held to a relative error
of 2e-3 of the output scale (operand rounding 2^-11); pure fp32 kernels to 1e-5..1e-4."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def h16(t):  # what the kernel sees: fp16-rounded operand
    return t.half().float()


@pytest.mark.parametrize("m,k,n,act", [(1, 1024, 1024, "none"), (8, 1024, 4096, "relu"), (8, 4096, 1024, "none"),
                                       (16, 320, 1024, "silu"), (32, 192, 80, "none"), (33, 512, 80, "none"), (32, 4096, 1024, "none"), (27, 2560, 512, "relu"),
                                       (300, 1024, 3072, "none"), (1000, 256, 1024, "gelu"), (5504, 1024, 256, "none"),
                                       (129, 80, 512, "mish"), (2000, 100, 18, "tanh"), (640, 512, 4097, "none")])
def test_linear(m, k, n, act):
    from astts import ops

    g = torch.Generator().manual_seed(m * 7 + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    pw = ops.PackedWeight(w, b)
    y = ops.linear(x.to(DEV), pw, act=act, residual=res.to(DEV), alpha=0.5)
    ref = F.linear(h16(x), h16(w), b)
    ref = {"none": lambda t: t, "relu": F.relu, "silu": F.silu, "gelu": F.gelu, "mish": F.mish, "tanh": torch.tanh}[act](ref)
    ref = ref * 0.5 + res
    assert rel_err(y, ref) < 2e-4          # vs the fp16-rounded-operand reference: only accumulation order differs
    full = F.linear(x, w, b)
    full = {"none": lambda t: t, "relu": F.relu, "silu": F.silu, "gelu": F.gelu, "mish": F.mish, "tanh": torch.tanh}[act](full) * 0.5 + res
    assert rel_err(y, full) < 3e-3         # vs full fp32: operand rounding


@pytest.mark.parametrize("b,t,cin,cout,k,stride,dil", [(2, 100, 80, 512, 7, 1, 1), (3, 257, 256, 256, 3, 1, 1),
                                                       (2, 300, 128, 128, 11, 1, 5), (2, 301, 256, 256, 3, 2, 1),
                                                       (1, 130, 512, 512, 3, 1, 3), (2, 513, 18, 256, 16, 8, 1),
                                                       (2, 200, 18, 128, 1, 1, 1), (2, 640, 128, 18, 7, 1, 1)])
def test_conv1d(b, t, cin, cout, k, stride, dil):
    from astts import ops

    g = torch.Generator().manual_seed(k * 100 + cin)
    x = torch.randn(b, t, cin, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(cout, generator=g)
    pad = {16: 4}.get(k, dil * (k - 1) // 2)
    pw = ops.PackedWeight.from_conv1d(w, bias)
    y = ops.conv1d(x.to(DEV), pw, stride=stride, dil=dil, pad=pad, act="leaky", slope=0.1)
    ref = F.leaky_relu(F.conv1d(h16(x).transpose(1, 2), h16(w), bias, stride=stride, dilation=dil, padding=pad), 0.1).transpose(1, 2)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-4


@pytest.mark.parametrize("b,t,cin,cout,s", [(2, 50, 512, 256, 8), (1, 129, 256, 128, 8), (2, 77, 256, 256, 2)])
def test_conv_transpose1d(b, t, cin, cout, s):
    from astts import ops

    g = torch.Generator().manual_seed(s)
    x = torch.randn(b, t, cin, generator=g)
    w = torch.randn(cin, cout, 2 * s, generator=g) / math.sqrt(cin * 2)
    bias = torch.randn(cout, generator=g)
    pw = ops.PackedWeight.from_conv_transpose1d(w, bias, s)
    y = ops.conv_transpose1d(x.to(DEV), pw, padding=s // 2)
    ref = F.conv_transpose1d(h16(x).transpose(1, 2), h16(w), bias, stride=s, padding=s // 2).transpose(1, 2)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-4


def test_layernorm_groupnorm():
    from astts import ops

    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 37, 1024, generator=g) * 3 + 1
    ga, be = torch.randn(1024, generator=g), torch.randn(1024, generator=g)
    y = ops.layernorm(x.to(DEV), ga.to(DEV), be.to(DEV), 1e-5)
    assert rel_err(y, F.layer_norm(x, (1024,), ga, be, 1e-5)) < 2e-5
    # GroupNorm(8, 256) + Mish over valid rows only, + per-(b,c) add, masked rows -> 0
    x = torch.randn(3, 150, 256, generator=g) * 2 - 0.5
    ga, be = torch.randn(256, generator=g), torch.randn(256, generator=g)
    add = torch.randn(3, 256, generator=g)
    lens = torch.tensor([150, 97, 1], dtype=torch.int32)
    y = ops.groupnorm(x.to(DEV), ga.to(DEV), be.to(DEV), 8, 1e-5, lens=lens.to(DEV), mish=True, add_bc=add.to(DEV)).cpu()
    for i, L in enumerate(lens.tolist()):
        ref = F.mish(F.group_norm(x[i:i + 1, :L].transpose(1, 2), 8, ga, be, 1e-5)).transpose(1, 2)[0] + add[i]
        assert rel_err(y[i, :L], ref) < 5e-5
        assert float(y[i, L:].abs().max()) == 0.0 if L < 150 else True
    x = torch.randn(2, 70, 80, generator=g)
    ga, be = torch.randn(80, generator=g), torch.randn(80, generator=g)
    y = ops.groupnorm(x.to(DEV), ga.to(DEV), be.to(DEV), 1, 1e-5)
    assert rel_err(y, F.group_norm(x.transpose(1, 2), 1, ga, be, 1e-5).transpose(1, 2)) < 5e-5


def test_elementwise_family():
    from astts import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 33, 128, generator=g)
    z = torch.randn(4, 33, 128, generator=g)
    al = torch.rand(128, generator=g) + 0.1
    xd, zd = x.to(DEV), z.to(DEV)
    assert rel_err(ops.elementwise(ops.EL_SNAKE, xd, p0=al.to(DEV)), x + torch.sin(al * x) ** 2 / (al + 1e-9)) < 1e-5
    assert rel_err(ops.elementwise(ops.EL_LEAKY, xd, s=0.1), F.leaky_relu(x, 0.1)) < 1e-6
    assert rel_err(ops.elementwise(ops.EL_ADD, xd, z=zd[:2].contiguous(), s=1 / 3), x + z[:2] / 3) < 1e-6
    lens = torch.tensor([20, 33], dtype=torch.int32)
    m = (torch.arange(33)[None, :] < lens[:, None]).float()[..., None]
    assert rel_err(ops.elementwise(ops.EL_MUL_ROWMASK, xd, lens=lens.to(DEV)), x * m) < 1e-7
    bc = torch.randn(2, 128, generator=g)
    assert rel_err(ops.elementwise(ops.EL_ADD_BC, xd, p0=bc.to(DEV)), x + bc[:, None, :]) < 1e-6
    assert rel_err(ops.elementwise(ops.EL_CFG_EULER, xd, z=zd, s=0.07, s2=0.7), x + 0.07 * (1.7 * z[:2] - 0.7 * z[2:])) < 1e-6
    assert rel_err(ops.elementwise(ops.EL_MISH, xd), F.mish(x)) < 1e-5
    assert rel_err(ops.elementwise(ops.EL_ELU, xd), F.elu(x)) < 1e-5
    assert rel_err(ops.elementwise(ops.EL_CLAMP, xd, s=0.99), x.clamp(-0.99, 0.99)) < 1e-7


def test_embedding_interp_time():
    from astts import ops

    g = torch.Generator().manual_seed(2)
    tab = torch.randn(4096, 512, generator=g)
    ids = torch.randint(0, 4096, (3, 50), generator=g)
    assert rel_err(ops.embedding(tab.to(DEV), ids.to(DEV)), tab[ids]) == 0.0
    x = torch.randn(2, 150, 80, generator=g)
    for t_out in (258, 430, 75, 150):
        ref = F.interpolate(x.transpose(1, 2), size=t_out, mode="linear").transpose(1, 2)
        assert rel_err(ops.interp_linear(x.to(DEV), t_out), ref) < 1e-5
    t = torch.tensor([0.0, 0.3, 0.97, 1.0])
    half = 160
    emb = 1000.0 * t[:, None] * torch.exp(torch.arange(half).float() * -(math.log(10000.0) / (half - 1)))[None, :]
    ref = torch.cat([emb.sin(), emb.cos()], dim=-1)
    assert float((ops.time_embedding(t.to(DEV), 320).cpu() - ref).abs().max()) < 2e-3  # fp32 sin/cos of args up to 1000


def _relpos_ref(q, k, v, pos_table, pos_center, bu, bv, heads, lens, q_pos0, causal):
    b, tq, _ = q.shape
    tk = k.shape[1]
    qh = q.view(b, tq, heads, 64).transpose(1, 2)
    kh = k.view(b, tk, heads, 64).transpose(1, 2)
    vh = v.view(b, tk, heads, 64).transpose(1, 2)
    i = torch.arange(tq)[:, None] + q_pos0
    j = torch.arange(tk)[None, :]
    p = pos_table[(i - j) + pos_center].view(tq, tk, heads, 64).permute(2, 0, 1, 3)  # [h, tq, tk, 64]
    ac = torch.einsum("bhid,bhjd->bhij", qh + bu.view(1, heads, 1, 64), kh)
    bd = torch.einsum("bhid,hijd->bhij", qh + bv.view(1, heads, 1, 64), p)
    s = (ac + bd) / 8.0
    mask = (j[None] < lens[:, None, None])
    if causal:
        mask = mask & (j <= i)[None]
    s = s.masked_fill(~mask[:, None], float("-inf"))
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), vh).transpose(1, 2).reshape(b, tq, heads * 64)


@pytest.mark.parametrize("b,heads,t,causal", [(2, 4, 50, False), (3, 16, 185, True), (2, 8, 400, False)])
def test_attn_relpos_prefill(b, heads, t, causal):
    from astts import ops

    g = torch.Generator().manual_seed(t)
    hd = heads * 64
    qkv = torch.randn(b, t, 3 * hd, generator=g)
    center = 700
    pos = torch.randn(2 * center + 1, hd, generator=g) * 0.5
    bu, bv = torch.randn(hd, generator=g) * 0.3, torch.randn(hd, generator=g) * 0.3
    lens = torch.tensor([t, max(1, t - 13), max(1, t // 2)][:b], dtype=torch.int32)
    qd = qkv.to(DEV)
    out = ops.attn_relpos(qd[..., :hd], qd[..., hd:2 * hd], qd[..., 2 * hd:], pos.to(DEV), bu.to(DEV), bv.to(DEV), heads,
                          lens=lens.to(DEV), q_pos0=0, pos_center=center, causal=causal).cpu()
    ref = _relpos_ref(qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:], pos, center, bu, bv, heads, lens, 0, causal)
    for i, L in enumerate(lens.tolist()):
        assert rel_err(out[i, :L], ref[i, :L]) < 4e-3        # fp16 MFMA operands (q+u, q+v, k, v, position rows), fp32 accumulate


@pytest.mark.parametrize("tq,tk,q_pos0,causal", [(40, 140, 100, True), (130, 130, 0, True), (97, 1000, 903, True), (600, 600, 0, False)])
def test_attn_relpos_prefill_chunks_and_long_sequences(tq, tk, q_pos0, causal):
    """Query chunks at an offset (absolute positions q_pos0 + i), sequences spanning many key tiles, left padding."""
    from astts import ops

    g = torch.Generator().manual_seed(tq + tk)
    b, heads = 2, 2
    hd = heads * 64
    q = torch.randn(b, tq, hd, generator=g)
    k = torch.randn(b, tk, hd, generator=g)
    v = torch.randn(b, tk, hd, generator=g)
    center = 1300
    pos = torch.randn(2 * center + 1, hd, generator=g) * 0.5
    bu, bv = torch.randn(hd, generator=g) * 0.3, torch.randn(hd, generator=g) * 0.3
    lens = torch.tensor([tk, tk - 7], dtype=torch.int32)
    out = ops.attn_relpos(q.to(DEV), k.to(DEV), v.to(DEV), pos.to(DEV), bu.to(DEV), bv.to(DEV), heads, lens=lens.to(DEV),
                          q_pos0=q_pos0, pos_center=center, causal=causal).cpu()
    ref = _relpos_ref(q, k, v, pos, center, bu, bv, heads, lens, q_pos0, causal)
    for i in range(b):
        rows = slice(0, tq) if causal else slice(0, min(tq, int(lens[i])))
        assert rel_err(out[i, rows], ref[i, rows]) < 4e-3
    # fp16 K/V and position table (the LM's cache layout) give the same answer as their fp32 values
    out16 = ops.attn_relpos(q.to(DEV), k.half().to(DEV), v.half().to(DEV), pos.half().to(DEV), bu.to(DEV), bv.to(DEV), heads,
                            lens=lens.to(DEV), q_pos0=q_pos0, pos_center=center, causal=causal).cpu()
    assert rel_err(out16, out) < 2e-3


def test_attn_relpos_decode_matches_prefill_row():
    from astts import ops

    g = torch.Generator().manual_seed(9)
    b, heads, tk = 8, 16, 333
    hd = heads * 64
    kc = torch.randn(b, 512, hd, generator=g)      # KV cache with capacity 512, 333 valid
    vc = torch.randn(b, 512, hd, generator=g)
    q = torch.randn(b, 1, hd, generator=g)
    center = 700
    pos = torch.randn(2 * center + 1, hd, generator=g) * 0.5
    bu, bv = torch.randn(hd, generator=g) * 0.3, torch.randn(hd, generator=g) * 0.3
    lens = torch.full((b,), tk, dtype=torch.int32)
    out = ops.attn_relpos(q.to(DEV), kc.to(DEV), vc.to(DEV), pos.to(DEV), bu.to(DEV), bv.to(DEV), heads,
                          lens=lens.to(DEV), q_pos0=tk - 1, pos_center=center, causal=False).cpu()
    ref = _relpos_ref(q, kc[:, :tk], vc[:, :tk], pos, center, bu, bv, heads, lens, tk - 1, False)
    assert rel_err(out, ref) < 1e-4


@pytest.mark.parametrize("b,heads,tk", [(128, 16, 333), (40, 16, 77), (64, 4, 1030), (256, 8, 130), (33, 12, 5)])
def test_attn_relpos_rows_wide_decode_batches(b, heads, tk):
    """The wide decode batches' attention (attn_relpos_rows: more than 32 rows over an fp16 time-major cache and an fp16 position table,
    one pass over the keys) against the fp32 definition on the fp16-rounded cache, with left-padded rows (key_start) -- and a row's
    result does not depend on the rows beside it (a 32-row launch takes the (row, head) kernel: equal to rounding; a 40-row sub-batch
    takes this kernel: equal bits)."""
    from astts import ops

    g = torch.Generator().manual_seed(b + tk)
    hd = heads * 64
    cap = tk + 7
    kc = (torch.randn(cap, b, hd, generator=g)).half()            # time-major cache [T, B, 2 * hd]: K | V interleaved per row
    vc = (torch.randn(cap, b, hd, generator=g)).half()
    kv = torch.cat([kc, vc], dim=2).contiguous().to(DEV)
    q = torch.randn(1, b, hd, generator=g)
    center = 1100
    pos = (torch.randn(2 * center + 1, hd, generator=g) * 0.5).half()
    bu, bv = torch.randn(hd, generator=g) * 0.3, torch.randn(hd, generator=g) * 0.3
    ks = torch.randint(0, max(tk // 2, 1), (b,), generator=g).to(torch.int32)
    ks[0] = 0
    ks[-1] = tk - 1                                                # a row with a single valid key
    kd, vd = kv[:tk, :, :hd], kv[:tk, :, hd:]
    out = ops.attn_relpos(q.to(DEV), kd, vd, pos.to(DEV), bu.to(DEV), bv.to(DEV), heads, q_pos0=tk - 1, pos_center=center,
                          time_major=True, key_start=ks.to(DEV))
    assert out.shape == (1, b, hd) and bool(torch.isfinite(out).all())
    for r in (0, 1, b // 2, b - 1):
        k0 = int(ks[r])
        lens = torch.tensor([tk - k0], dtype=torch.int32)
        ref = _relpos_ref(q[:, r:r + 1].transpose(0, 1), kc[k0:tk, r:r + 1].float().transpose(0, 1), vc[k0:tk, r:r + 1].float().transpose(0, 1),
                          pos.float(), center, bu, bv, heads, lens, tk - 1 - k0, False)
        assert rel_err(out[0, r].cpu(), ref[0, 0]) < 2e-5, r
    n2 = 32
    small = ops.attn_relpos(q[:, :n2].contiguous().to(DEV), kv[:tk, :n2, :hd], kv[:tk, :n2, hd:], pos.to(DEV), bu.to(DEV), bv.to(DEV), heads,
                            q_pos0=tk - 1, pos_center=center, time_major=True, key_start=ks[:n2].contiguous().to(DEV))
    assert rel_err(small, out[:, :n2]) < 2e-5
    if b >= 48:
        n3 = 40
        sub = ops.attn_relpos(q[:, :n3].contiguous().to(DEV), kv[:tk, :n3, :hd], kv[:tk, :n3, hd:], pos.to(DEV), bu.to(DEV), bv.to(DEV), heads,
                              q_pos0=tk - 1, pos_center=center, time_major=True, key_start=ks[:n3].contiguous().to(DEV))
        assert torch.equal(sub, out[:, :n3])


@pytest.mark.parametrize("b,heads,t", [(2, 8, 344), (3, 8, 688), (1, 2, 31), (2, 4, 129)])
def test_attn_mha_flash(b, heads, t):
    from astts import ops

    g = torch.Generator().manual_seed(t)
    hd = heads * 64
    qkv = torch.randn(b, t, 3 * hd, generator=g)
    lens = torch.tensor([t, max(1, t - 40), max(1, t // 3)][:b], dtype=torch.int32)
    qd = qkv.to(DEV)
    out = ops.attn_mha(qd[..., :hd], qd[..., hd:2 * hd], qd[..., 2 * hd:], heads, lens=lens.to(DEV)).cpu()
    q, k, v = (qkv[..., i * hd:(i + 1) * hd].view(b, t, heads, 64).transpose(1, 2) for i in range(3))
    mask = (torch.arange(t)[None, :] < lens[:, None])[:, None, None, :]
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(b, t, hd)
    for i, L in enumerate(lens.tolist()):
        assert rel_err(out[i, :L], ref[i, :L]) < 3e-3   # fp16 operands (Q, K, P, V), fp32 softmax/accumulate


def test_stft_istft_nsf():
    from astts import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 2560, generator=g)
    win = torch.hann_window(16, periodic=True)
    spec = torch.stft(x, 16, 4, 16, window=win, return_complex=True)       # [B, 9, F]
    ref = torch.cat([spec.real, spec.imag], dim=1).transpose(1, 2)         # [B, F, 18]
    y = ops.stft16(x.to(DEV))
    assert y.shape == ref.shape and rel_err(y, ref) < 1e-5
    pre = torch.randn(2, 641, 18, generator=g) * 0.5
    mag = torch.clip(torch.exp(pre[..., :9]), max=100.0)
    ph = torch.sin(pre[..., 9:])
    cplx = torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)).transpose(1, 2)
    refw = torch.istft(cplx, 16, 4, 16, window=win).clamp(-0.99, 0.99)
    w = ops.istft16(pre.to(DEV))
    assert w.shape == refw.shape and float((w.cpu() - refw).abs().max()) < 2e-5
    # NSF source: fp64 phase accumulation (oracle definition: cumsum in float64)
    b, tm, up, nh, sr = 2, 40, 256, 9, 22050.0
    f0 = torch.rand(b, tm, generator=g) * 300
    f0[:, 5:9] = 0.0                                                        # unvoiced stretch
    phase0 = (torch.rand(b, nh, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(b, tm * up, nh, generator=g)
    lw, lb = torch.randn(nh, generator=g) * 0.3, torch.randn(1, generator=g) * 0.1
    f0u = f0.double().repeat_interleave(up, dim=1)                          # nearest upsample
    harm = torch.arange(1, nh + 1).double()
    theta = 2 * math.pi * ((torch.cumsum(f0u / sr, dim=1)[..., None] * harm) % 1.0)
    sine = 0.1 * torch.sin(theta.float() + phase0[:, None, :])
    uv = (f0u > 10.0).float()[..., None]
    src = sine * uv + (uv * 0.003 + (1 - uv) * 0.1 / 3) * noise
    ref = torch.tanh(src @ lw + lb)
    out = ops.nsf_source(f0.to(DEV), phase0.to(DEV), noise.to(DEV), lw.to(DEV), lb.to(DEV), up, sr, 0.1, 0.003, 10.0)
    assert float((out.cpu() - ref).abs().max()) < 2e-5


def test_ras_sample_matches_definition():
    from astts import ops
    from oracle import synth as osyn

    g = torch.Generator().manual_seed(4)
    b, v = 16, 4097
    logits = torch.randn(b, v, generator=g) * 3
    hist = torch.randint(0, 4096, (b, 64), generator=g, dtype=torch.int32)
    u = torch.rand(b, 2, generator=g)
    for hist_len in (0, 3, 40):
        for ignore in (True, False):
            if hist_len >= 3:
                # force repetition in row 0: make its most likely token fill the window
                top = int(logits[0].argmax())
                hist[0, :hist_len] = top
            out = ops.ras_sample(logits.to(DEV), hist.to(DEV), hist_len, u.to(DEV), 25, 0.8, 10, 0.1, 4096, ignore).cpu()
            ref = osyn.ras_sample(logits, hist[:, :hist_len], u, 25, 0.8, 10, 0.1, 4096, ignore)
            assert out.tolist() == ref.tolist()
    # degenerate distributions: fewer non-zero probabilities than top_k (ties at p == 0 -> the general path), and
    # exact ties among the top entries (order by id)
    lg2 = torch.full((4, v), float("-inf"))
    lg2[0, [5, 77, 4000]] = torch.tensor([1.0, 1.0, 0.5])
    lg2[1, 123] = 0.0
    lg2[2, :40] = 2.0                       # 40-way exact tie
    lg2[3] = torch.randn(v, generator=g)
    lg2[3, 1000:1030] = lg2[3].max() + 1.0  # 30-way tie at the top
    for uu in (torch.tensor([[0.0, 0.0]] * 4), torch.tensor([[0.999, 0.5]] * 4), torch.rand(4, 2, generator=g)):
        out = ops.ras_sample(lg2.to(DEV), hist[:4].contiguous().to(DEV), 0, uu.to(DEV), 25, 0.8, 10, 0.1, 4096, False).cpu()
        ref = osyn.ras_sample(lg2, hist[:4, :0], uu, 25, 0.8, 10, 0.1, 4096, False)
        assert out.tolist() == ref.tolist()


def test_ras_sample_reject_policy_matches_definition():
    """eos_policy "reject" (upstream's re-draw until the token is not EOS, in closed form: oracle.synth.ras_sample): EOS inside the
    nucleus, EOS holding most of the mass (the nucleus is EOS alone: the draw falls through to the distribution without EOS),
    repeated tokens inside the nucleus, both uniforms across their range; never returns EOS; and equals the "mask" policy's answer
    only by accident (the policies are different distributions)."""
    from astts import ops
    from oracle import synth as osyn

    g = torch.Generator().manual_seed(21)
    b, v, eos = 16, 4097, 4096
    logits = torch.randn(b, v, generator=g) * 3
    logits[0, eos] = logits[0].max() + 1.0                  # EOS is the most likely token
    logits[1, eos] = logits[1].max() + 12.0                 # ... and nearly everything: the nucleus holds EOS only
    logits[2, eos] = logits[2].topk(5).values[-1]           # EOS in the middle of the nucleus
    logits[3] = -30.0
    logits[3, [eos, 7]] = torch.tensor([5.0, 0.0])          # EOS + one token
    hist = torch.randint(0, 4096, (b, 64), generator=g, dtype=torch.int32)
    differs = 0
    for hist_len in (0, 5, 40):
        if hist_len:
            for r in range(0, b, 2):                        # repeated tokens inside the nucleus of every other row
                top = logits[r].topk(3).indices
                hist[r, hist_len - 1] = int(top[0]) if int(top[0]) != eos else int(top[1])
                hist[r, hist_len - 2] = int(top[2]) if int(top[2]) != eos else int(top[1])
        for u_fix in (None, (0.0, 0.0), (0.999999, 0.999999), (0.5, 0.0)):
            u = torch.rand(b, 2, generator=g)
            if u_fix is not None:
                u[:, 0], u[:, 1] = u_fix
            out = ops.ras_sample(logits.to(DEV), hist.to(DEV), hist_len, u.to(DEV), 25, 0.8, 10, 0.1, eos, True, eos_policy="reject").cpu()
            ref = osyn.ras_sample(logits, hist[:, :hist_len], u, 25, 0.8, 10, 0.1, eos, True, "reject")
            assert out.tolist() == ref.tolist(), (hist_len, u_fix)
            assert int(out.max()) < eos
            masked = ops.ras_sample(logits.to(DEV), hist.to(DEV), hist_len, u.to(DEV), 25, 0.8, 10, 0.1, eos, True).cpu()
            differs += int((masked != out).sum())
            # outside the window the policy bit changes nothing
            free_r = ops.ras_sample(logits.to(DEV), hist.to(DEV), hist_len, u.to(DEV), 25, 0.8, 10, 0.1, eos, False, eos_policy="reject").cpu()
            free_m = ops.ras_sample(logits.to(DEV), hist.to(DEV), hist_len, u.to(DEV), 25, 0.8, 10, 0.1, eos, False).cpu()
            assert free_r.tolist() == free_m.tolist()
    assert differs > 0


def test_ras_sample_repetition_path_every_row():
    """The repetition branch (inverse CDF over the FULL distribution in id order, sequential fp32 running sum) for every row
    (win = 0 makes the repetition test pass trivially), with u2 across the range, u2 -> 1 (the sum may never exceed the target:
    the last token with p > 0) and degenerate distributions; vocabulary sizes around the 16-element chunking of the scan."""
    from astts import ops
    from oracle import synth as osyn

    g = torch.Generator().manual_seed(11)
    for v in (4097, 4096, 1000, 17):
        b = 12
        logits = torch.randn(b, v, generator=g) * 3
        logits[1] = float("-inf")
        logits[1, [3, v - 1]] = 0.0                     # two tokens, the second one last
        logits[2] = float("-inf")
        logits[2, v // 2] = 1.0                         # a single token
        logits[3, v - 5:] = float("-inf")               # zero-probability tail
        hist = torch.randint(0, v - 1, (b, 8), generator=g, dtype=torch.int32)
        for u2 in (None, 0.0, 0.5, 0.999999, 1.0):
            u = torch.rand(b, 2, generator=g)
            if u2 is not None:
                u[:, 1] = u2
            out = ops.ras_sample(logits.to(DEV), hist.to(DEV), 8, u.to(DEV), 25, 0.8, 0, 0.1, v - 1, False).cpu()
            ref = osyn.ras_sample(logits, hist, u, 25, 0.8, 0, 0.1, v - 1, False)
            # u2 == 1.0: whether a running sum of ~1.0 "exceeds" the target is decided by the last ulp of the softmax
            # normaliser (the kernel and torch sum in different orders); only the rows whose probabilities are exact
            # (0.5 + 0.5, 1.0: the sum never exceeds 1.0 -> last token with p > 0) are comparable there
            rows = [1, 2] if u2 == 1.0 else list(range(b))
            assert out[rows].tolist() == ref[rows].tolist(), (v, u2)


def test_gemm_rejects_in_place_output():
    """Every GEMM workgroup reads whole input rows while others store their output tiles: out == x would race
    (it only shows under contention), so the ABI refuses it."""
    from astts import _lib, ops

    x = torch.randn(256, 256, device=DEV)
    pw = ops.PackedWeight(torch.randn(256, 256) / 16, None)
    with pytest.raises(_lib.AsttsError, match="aliases"):
        ops.gemm(x, pw, out=x)


@pytest.mark.parametrize("m,k,n,act,use_res", [(8, 4096, 1024, "none", True), (24, 4096, 1024, "none", True), (32, 4096, 1024, "relu", False),
                                               (5, 2048, 512, "relu", False), (8, 1024, 1024, "none", True), (16, 4096, 2048, "none", False)])
def test_gemm_fused_split_k_and_plain(m, k, n, act, use_res):
    """Decode-sized GEMM; K >= 2048 onto <= 2048 columns takes the split-K path (4 K slices per column block, the last
    block to arrive reduces in slice order): checked against the fp16-operand reference, for run-to-run bit
    reproducibility, and over several launches sharing the self-resetting workspace."""
    from astts import ops

    g = torch.Generator().manual_seed(m * 13 + n)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    pw = ops.PackedWeight(w, b)
    outs = []
    for rep in range(3):
        x = torch.randn(m, k, generator=g)
        res = torch.randn(m, n, generator=g) if use_res else None
        xd = x.to(DEV)
        rd = None if res is None else res.to(DEV)
        y = ops.gemm_fused(xd, pw, m, act=act, residual=rd)
        y2 = ops.gemm_fused(xd, pw, m, act=act, residual=rd)
        assert torch.equal(y, y2)                       # same bits whichever block arrives last
        ref = F.linear(h16(x), h16(w), b)
        ref = F.relu(ref) if act == "relu" else ref
        if res is not None:
            ref = ref + res
        assert rel_err(y, ref) < 2e-4, rep
        outs.append(y)


@pytest.mark.parametrize("m,k,n,act,x16,out16,use_res", [(128, 1024, 1024, "none", True, False, True), (128, 1024, 2048, "none", True, True, False),
                                                         (128, 1024, 4096, "relu", True, True, False), (128, 4096, 1024, "none", True, False, True),
                                                         (128, 1024, 1024, "none", False, False, True), (128, 1024, 4097, "none", True, False, False),
                                                         (33, 1024, 1024, "none", True, False, True), (256, 1024, 4096, "relu", True, True, False),
                                                         (77, 2048, 528, "gelu", False, False, True), (64, 1024, 1024, "none", False, False, False),
                                                         (200, 576, 100, "silu", True, False, True), (40, 6144, 64, "none", True, False, False),
                                                         (100, 2560, 1000, "none", False, True, True)])
def test_gemm_rows_matches_definition_rows_are_independent(m, k, n, act, x16, out16, use_res):
    """astts_op_gemm_rows (the wide decode engine's projections: 33 .. 256 rows, one memory round trip per workgroup) against the
    fp16-operand reference over its tile shapes (wide / narrow outputs, K lines per wave 2 / 4 / 8 and a ragged last pass, fp32 and fp16
    activations and outputs, a column count that is not a multiple of the tile) -- and a row's result does not depend on the rows
    beside it: the first 40 rows run as a launch of their own give the same bits."""
    from astts import ops

    g = torch.Generator().manual_seed(m * 31 + n + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g) if use_res else None
    pw = ops.PackedWeight(w, b)
    xd = x.to(DEV).half() if x16 else x.to(DEV)
    rd = None if res is None else res.to(DEV)
    od = torch.float16 if out16 else torch.float32
    y = ops.gemm_rows(xd, pw, act=act, residual=rd, out_dtype=od)
    assert y.dtype == od and y.shape == (m, n)
    ref = F.linear(h16(x), h16(w), b)
    ref = {"none": lambda t: t, "relu": F.relu, "silu": F.silu, "gelu": F.gelu}[act](ref)
    if res is not None:
        ref = ref + res
    assert rel_err(y.float(), ref) < (1.5e-3 if out16 else 2e-4)
    assert torch.equal(y, ops.gemm_rows(xd, pw, act=act, residual=rd, out_dtype=od))            # run to run
    sub = min(40, m)
    y2 = ops.gemm_rows(xd[:sub].contiguous(), pw, act=act, residual=None if rd is None else rd[:sub].contiguous(), out_dtype=od)
    assert torch.equal(y2, y[:sub])
    # a slice of the weight rows (the engine's q and k|v halves of one packed projection) into a strided destination
    if n >= 512:
        big = torch.zeros((m, n + 64), dtype=od, device=DEV)
        ops.gemm_rows(xd, pw, act=act, out=big[:, 32:32 + 256], n=256, row0=128)
        want = ops.gemm_rows(xd, pw, act=act, out_dtype=od)[:, 128:384]
        assert torch.equal(big[:, 32:288], want) and float(big[:, :32].abs().max()) == 0.0 and float(big[:, 288:].abs().max()) == 0.0


@pytest.mark.parametrize("m,k,n,n_split,x16", [(128, 1024, 3072, 1024, True), (256, 512, 1536, 512, True), (50, 1024, 2048, 1024, False)])
def test_gemm_rows_split_output(m, k, n, n_split, x16):
    """astts_op_gemm_rows with the columns >= n_split written as fp16 into a strided second destination (the KV-cache row of the wide
    engine's q | k | v launch): both halves against the fp16-operand product, nothing written beside them, and bit-equal for a
    sub-batch of the rows."""
    from astts import ops

    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    pw = ops.PackedWeight(w, b)
    xd = x.to(DEV).half() if x16 else x.to(DEV)
    cache = torch.zeros((m, 2 * (n - n_split)), dtype=torch.float16, device=DEV)
    y = ops.gemm_rows(xd, pw, out2=cache[:, 16:16 + n - n_split], n_split=n_split)
    ref = F.linear(h16(x), h16(w), b)
    assert y.shape == (m, n_split) and rel_err(y, ref[:, :n_split]) < 2e-4
    assert rel_err(cache[:, 16:16 + n - n_split].float(), ref[:, n_split:]) < 1.5e-3
    assert float(cache[:, :16].abs().max()) == 0.0 and float(cache[:, 16 + n - n_split:].abs().max()) == 0.0
    sub = min(40, m)
    c2 = torch.zeros((sub, n - n_split), dtype=torch.float16, device=DEV)
    y2 = ops.gemm_rows(xd[:sub].contiguous(), pw, out2=c2, n_split=n_split)
    assert torch.equal(y2, y[:sub]) and torch.equal(c2, cache[:sub, 16:16 + n - n_split])


def test_gemm_fused_layernorm_gather_split_output():
    """The fusions of the LM decode step: embedding-row gather, LayerNorm prologue, K|V half of the output written as
    fp16 into a strided destination (a KV-cache row)."""
    from astts import ops

    g = torch.Generator().manual_seed(5)
    m, k, n = 8, 1024, 3072
    table = torch.randn(50, k, generator=g)
    ids = torch.randint(0, 50, (m,), generator=g).to(torch.int32)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g)
    ga, be = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    pw = ops.PackedWeight(w, b)
    cache = torch.zeros(3, m, 2 * 1024, dtype=torch.float16, device=DEV)
    q = ops.gemm_fused(table.to(DEV), pw, m, gather=ids.to(DEV), ln=(ga.to(DEV), be.to(DEV)), ln_eps=1e-5, out2=cache[1], n_split=1024)
    xn = F.layer_norm(table[ids.long()], (k,), ga, be, 1e-5)
    ref = F.linear(h16(xn), h16(w), b)
    assert rel_err(q, ref[:, :1024]) < 2e-3                 # LayerNorm output rounded to fp16 before the MFMA
    assert rel_err(cache[1].float(), ref[:, 1024:]) < 3e-3
    assert float(cache[0].abs().max()) == 0.0 and float(cache[2].abs().max()) == 0.0


@pytest.mark.parametrize("m,k", [(333, 512), (5504, 1024), (64, 256), (31, 64)])
def test_linear_with_fused_residual_and_layernorm(m, k):
    """astts_op_gemm_ln: out = x @ w^T + b + residual (fp32) and LayerNorm(out) (fp16) from one launch."""
    from astts import ops

    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).half()
    w = torch.randn(256, k, generator=g) / math.sqrt(k)
    b = torch.randn(256, generator=g)
    res = torch.randn(m, 256, generator=g) * 3.0 + 0.7
    ga, be = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.2
    pw = ops.PackedWeight(w, b)
    out, ln = ops.linear_ln(x.to(DEV), pw, res.to(DEV), (ga.to(DEV), be.to(DEV)), 1e-5)
    ref = F.linear(x.float(), h16(w), b) + res
    assert rel_err(out, ref) < 2e-4
    ref_ln = F.layer_norm(out.cpu(), (256,), ga, be, 1e-5)            # LayerNorm of the kernel's own fp32 result
    assert float((ln.float().cpu() - ref_ln).abs().max()) < 4e-3     # fp16 output rounding
    # the unfused pair of launches gives the same fp32 result bit for bit and the same LayerNorm up to fp16 rounding
    out2 = ops.linear(x.to(DEV), pw, residual=res.to(DEV))
    assert rel_err(out2, out) < 1e-6


@pytest.mark.parametrize("sr_in,sr_out,n", [(16000, 22050, 24000), (22050, 16000, 30001), (16000, 24000, 5000), (44100, 16000, 44100)])
def test_resampler_kernel_matches_host_definition(sr_in, sr_out, n):
    """astts_op_resample_poly vs the host form of the same Hann-windowed sinc table (astts.audio.resample)."""
    from astts import audio

    g = torch.Generator().manual_seed(n)
    t = torch.arange(n) / sr_in
    x = (0.4 * torch.sin(2 * math.pi * 440.0 * t) + 0.1 * torch.randn(n, generator=g))[None, :]
    ref = audio.resample(x, sr_in, sr_out)
    out = audio.resample(x.to(DEV), sr_in, sr_out)
    assert out.shape == ref.shape and out.is_cuda
    assert float((out.cpu() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("sr,n_fft,hop,n_mels,fmin,fmax,n", [(22050, 1024, 256, 80, 0.0, 8000.0, 66150), (16000, 400, 160, 80, 20.0, 7600.0, 24000),
                                                            (16000, 400, 320, 128, 0.0, 8000.0, 16123), (24000, 1024, 256, 80, 0.0, 8000.0, 9000)])
def test_mel_spectrogram_kernel_matches_host_definition(sr, n_fft, hop, n_mels, fmin, fmax, n):
    """astts_op_mel_spectrogram (direct fp32 DFT per frame) vs torch.stft on the host: reflect padding, Hann window,
    magnitude, Slaney mel, log floor 1e-5.  Tolerance: 2e-4 on the log-mel (fp32 summation order of 1024-point sums)."""
    from astts import audio

    g = torch.Generator().manual_seed(n)
    t = torch.arange(n) / sr
    x = torch.stack([0.3 * torch.sin(2 * math.pi * 220.0 * t) + 0.05 * torch.randn(n, generator=g),
                     0.5 * torch.sin(2 * math.pi * 1760.0 * t) * torch.exp(-3.0 * t) + 0.01 * torch.randn(n, generator=g)])
    ref = audio.mel_spectrogram(x, sr=sr, n_fft=n_fft, hop=hop, win=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
    out = audio.mel_spectrogram(x.to(DEV), sr=sr, n_fft=n_fft, hop=hop, win=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
    assert out.shape == ref.shape and out.is_cuda
    assert float((out.cpu() - ref).abs().max()) < 2e-4


def test_whisper_log_mel_kernel_matches_the_feature_extractor_fixture():
    """astts_op_whisper_log_mel (the 128-bin log-mel the reference's speech tokenizer takes: SURVEY.md a12 / 8f rank 3) against the
    committed WhisperFeatureExtractor output (tests/golden/synth_blocks.npz, made by transformers in the build container) and
    against the host form of the same definition; batch of two utterances whose maxima differ (the "max - 8" floor is per
    utterance), lengths that are / are not multiples of the hop."""
    from astts import audio

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth_blocks.npz"))
    wav = torch.from_numpy(fx["whisper.wav"])
    out = audio.whisper_log_mel(wav.to(DEV))
    ref = fx["whisper.features"]
    assert out.is_cuda and tuple(out.shape) == (1,) + ref.shape
    err = float(np.abs(out[0].cpu().numpy() - ref).max())
    print(f"whisper log-mel (HIP) vs WhisperFeatureExtractor: max abs diff {err:.2e}")
    assert err < 1e-4                                              # observed 2.1e-5 (the host torch.stft path: 2.3e-5)
    g = torch.Generator().manual_seed(8)
    for n in (16000, 16000 * 3 + 77, 4000):
        t = torch.arange(n) / 16000.0
        x = torch.stack([0.4 * torch.sin(2 * math.pi * 330.0 * t) + 0.02 * torch.randn(n, generator=g),
                         0.003 * torch.sin(2 * math.pi * 2500.0 * t) + 0.0005 * torch.randn(n, generator=g)])     # 40 dB quieter: its own floor
        host = audio.whisper_log_mel(x)
        dev = audio.whisper_log_mel(x.to(DEV)).cpu()
        assert dev.shape == host.shape == (2, 128, n // 160)
        assert float((dev - host).abs().max()) < 1e-4, n               # observed <= 3e-5
        assert float(dev[0].max() - dev[0].min()) <= 2.0 + 1e-5 and float(dev[1].max() - dev[1].min()) <= 2.0 + 1e-5    # floor at max - 8 -> (x + 4) / 4
        assert float(dev[0].max()) > float(dev[1].max()) + 0.5                  # each utterance has its own maximum


def test_kaldi_fbank_kernel_matches_the_fixture_and_the_host_definition():
    """astts_op_kaldi_fbank (the speaker network's input features) against the committed output of transformers'
    Kaldi-mimicking extractor and against the host (float64 FFT) form; both sample scales, a batch, lengths around the frame grid."""
    from astts import audio

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kaldi_fbank.npz"))
    wav = torch.from_numpy(fx["wav"])
    out = audio.kaldi_fbank(wav.to(DEV), scale=32768.0)
    assert out.is_cuda and tuple(out.shape) == (1,) + fx["features"].shape
    err = float(np.abs(out[0].cpu().numpy() - fx["features"]).max())
    print(f"kaldi fbank (HIP) vs SeamlessM4TFeatureExtractor: max abs diff {err:.2e}")
    assert err < 2e-4
    g = torch.Generator().manual_seed(12)
    for n in (400, 559, 560, 16000 * 2 + 3):
        t = torch.arange(n) / 16000.0
        x = torch.stack([0.4 * torch.sin(2 * math.pi * 330.0 * t) + 0.02 * torch.randn(n, generator=g) + 0.05,
                         0.01 * torch.sin(2 * math.pi * 2500.0 * t) + 0.001 * torch.randn(n, generator=g)])
        for scale in (1.0, 32768.0):
            host = audio.kaldi_fbank(x, scale=scale)
            dev = audio.kaldi_fbank(x.to(DEV), scale=scale).cpu()
            assert dev.shape == host.shape == (2, 1 + (n - 400) // 160, 80)
            assert float((dev - host).abs().max()) < 5e-4, (n, scale, float((dev - host).abs().max()))
    with pytest.raises(ValueError):
        audio.kaldi_fbank(torch.zeros(1, 300, device=DEV))


def test_new_entry_points_validate_arguments():
    """Argument errors come back as AsttsError with a message, before anything is launched."""
    import ctypes

    from astts import _lib, audio, ops

    lib = _lib.load()
    x = torch.randn(64, 128, device=DEV).half()
    pw = ops.PackedWeight(torch.randn(128, 128) / 11, None)            # n = 128: astts_op_gemm_ln needs 256
    res = torch.zeros(64, 128, device=DEV)
    g = torch.ones(128, device=DEV)
    with pytest.raises(AssertionError):
        ops.linear_ln(x, pw, res, (g, g))
    out = torch.empty(64, 128, device=DEV)
    ln = torch.empty(64, 128, dtype=torch.float16, device=DEV)
    rc = lib.astts_op_gemm_ln(x.data_ptr(), pw.data.data_ptr(), None, res.data_ptr(), out.data_ptr(), g.data_ptr(), g.data_ptr(), 1e-5,
                              ln.data_ptr(), 64, 128, 128, 128, 128, 128, 128, 128, _lib.stream_ptr())
    assert rc == _lib.ERR_UNSUPPORTED and b"n must be 256" in lib.astts_last_error_string()
    with pytest.raises(_lib.AsttsError, match="astts_op_mel_spectrogram"):
        audio.mel_spectrogram(torch.zeros(1, 100, device=DEV), n_fft=1024, hop=256, win=1024)
    rc = lib.astts_stream_spin(-5, _lib.stream_ptr())
    assert rc == _lib.ERR_INVALID
    rc = lib.astts_op_gemm_fused_ws(x.data_ptr(), None, None, None, ctypes.c_float(0.0), pw.data.data_ptr(), None, None, out.data_ptr(), None, 0,
                                    8, 128, 0, 128, 128, 128, 128, 0, 0, 0, ctypes.c_float(1.0), ctypes.c_float(0.1), ctypes.c_void_p(256), 16,
                                    _lib.stream_ptr())
    assert rc == _lib.ERR_WORKSPACE
    # astts_op_gemm_rows: K must be whole 64-element lines, the split output needs a second destination wide enough, operands aligned
    x16 = torch.randn(64, 128, device=DEV).half()
    y = torch.empty(64, 128, device=DEV)
    args = lambda k, n_split, out2, ldc2, lda: (x16.data_ptr(), 1, pw.data.data_ptr(), None, None, y.data_ptr(), 0, out2, 1, 64, 128, n_split, k, lda, 128, ldc2,
                                                 0, 0, _lib.stream_ptr())
    assert lib.astts_op_gemm_rows(*args(96, 0, None, 0, 128)) == _lib.ERR_UNSUPPORTED and b"multiple of 64" in lib.astts_last_error_string()
    assert lib.astts_op_gemm_rows(*args(128, 64, y.data_ptr(), 32, 128)) == _lib.ERR_INVALID and b"split output" in lib.astts_last_error_string()
    assert lib.astts_op_gemm_rows(*args(128, 0, None, 0, 100)) == _lib.ERR_INVALID
    assert lib.astts_op_gemm_rows(*args(128, 0, None, 0, 128)) == 0


@pytest.mark.parametrize("b,t,ragged", [(2, 70, True), (16, 344, False), (3, 352, True), (1, 33, False), (2, 1, False),
                                        (16, 375, False), (3, 384, True), (8, 353, True)])
def test_tfm_attn_fused_matches_definition_and_unfused_path(b, t, ragged):
    """astts_op_tfm_attn_fused (LayerNorm + q|k|v projection + masked MHA of a flow-estimator transformer block in one launch)
    against the fp32 definition and against the three-launch path it replaces (layernorm -> linear -> attn_mha) on the same
    folded weights.  Shapes: the benchmark's (16 sequences x 344 frames at 22.05 kHz, x 375 at 24 kHz), the largest supported T
    (384), ragged lengths, tiny T."""
    import torch.nn.functional as F

    from astts import ops
    from astts.synth.model import fold_layernorm

    heads, c = 8, 256
    g = torch.Generator().manual_seed(b * 1000 + t)
    x = torch.randn(b, t, c, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    w = torch.randn(3 * heads * 64, c, generator=g) / 16
    lens = torch.tensor([t] + [max(1, t - 7 * (i + 1)) for i in range(b - 1)]) if ragged else torch.full((b,), t)
    assert ops.tfm_attn_fused_supported(c, heads, t) and not ops.tfm_attn_fused_supported(c, heads, 385)
    wf, bf = fold_layernorm(w, torch.zeros(w.shape[0]), gamma, beta)
    pw = ops.PackedWeight(wf, bf)
    ld = lens.to(DEV, torch.int32)
    out = ops.tfm_attn_fused(x.to(DEV), pw, ops.tfm_pack_frag(pw), heads, lens=ld).float().cpu()
    # fp32 definition
    n = F.layer_norm(x, (c,), gamma, beta, 1e-5)
    qkv = n @ w.T
    q, k, v = (qkv[..., i * 512:(i + 1) * 512].view(b, t, heads, 64).transpose(1, 2) for i in range(3))
    mask = (torch.arange(t)[None, :] >= lens[:, None])[:, None, None, :]
    s = (q @ k.transpose(-1, -2) / 8.0).masked_fill(mask, float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b, t, 512)
    # the path it replaces
    ident = (torch.ones(c, device=DEV), torch.zeros(c, device=DEV))
    n16 = ops.layernorm(x.to(DEV), *ident, 1e-5, out_dtype=torch.float16)
    qkv16 = ops.linear(n16, pw, out_dtype=torch.float16)
    un = ops.attn_mha(qkv16[..., :512], qkv16[..., 512:1024], qkv16[..., 1024:], heads, lens=ld, out_dtype=torch.float16).float().cpu()
    for i in range(b):
        L = int(lens[i])
        scale = float(ref[i, :L].abs().max())
        e_ref = float((out[i, :L] - ref[i, :L]).abs().max()) / scale
        e_un = float((out[i, :L] - un[i, :L]).abs().max()) / scale
        assert e_ref < 4e-3 and e_un < 3e-3, (i, e_ref, e_un)
    assert bool(torch.isfinite(out).all())


@pytest.mark.parametrize("b,t,ragged", [(64, 344, False), (24, 375, True), (5, 384, True), (3, 33, True), (8, 1, False), (17, 353, True)])
def test_tfm_attn_fused_full_form_is_bit_identical_to_the_half_form(b, t, ragged, monkeypatch):
    """Beyond one round of (query half, head, sequence) workgroups a workgroup takes BOTH query halves of its (head, sequence): K and V are
    projected once, the second half's Q rows wait in the output tensor's rows.  Same arithmetic per row in the same order: the outputs
    of the two forms are equal bit for bit (so which form a batch takes is invisible in the results), over ragged lengths, odd and
    single chunk counts; the automatic choice (more than 16 sequences at 8 heads) equals both."""
    from astts import ops
    from astts.synth.model import fold_layernorm

    heads, c = 8, 256
    g = torch.Generator().manual_seed(b * 77 + t)
    x = (torch.randn(b, t, c, generator=g) * 2 + 0.3).to(DEV)
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    w = torch.randn(3 * heads * 64, c, generator=g) / 16
    lens = torch.tensor([t] + [max(1, t - 5 * (i + 1)) for i in range(b - 1)]) if ragged else torch.full((b,), t)
    wf, bf = fold_layernorm(w, torch.randn(w.shape[0], generator=g) * 0.1, gamma, beta)
    pw = ops.PackedWeight(wf, bf)
    frag = ops.tfm_pack_frag(pw)
    ld = lens.to(DEV, torch.int32)
    outs = {}
    for form in ("0", "1"):
        monkeypatch.setenv("ASTTS_TFM_ATTN_FULL", form)
        outs[form] = ops.tfm_attn_fused(x, pw, frag, heads, lens=ld)
        again = ops.tfm_attn_fused(x, pw, frag, heads, lens=ld)
        assert torch.equal(outs[form], again)
    monkeypatch.delenv("ASTTS_TFM_ATTN_FULL")
    auto = ops.tfm_attn_fused(x, pw, frag, heads, lens=ld)
    for i in range(b):
        L = int(lens[i])
        assert torch.equal(outs["0"][i, :L], outs["1"][i, :L]), i
        assert torch.equal(auto[i, :L], outs["0"][i, :L]), i
    assert bool(torch.isfinite(outs["1"].float()).all())


@pytest.mark.parametrize("m,hidden", [(5504, 1024), (37, 1024), (1, 256), (11008, 1024), (96, 512), (40000, 2048)])
def test_tfm_ffn_fused_matches_definition_and_unfused_path(m, hidden):
    """astts_op_tfm_ffn_fused (LayerNorm + Linear + exact-erf GELU + Linear + residual of a flow-estimator transformer block in one
    launch) against the fp64 definition and against the three launches it replaces on the same folded weights.  Rows: the
    benchmark's 16 x 344 and 16 x 688, a ragged last row block, a single row, several rounds of workgroups; two runs are bit-equal."""
    import torch.nn.functional as F

    from astts import ops
    from astts.synth.model import fold_layernorm

    c = 256
    g = torch.Generator().manual_seed(m + hidden)
    x = torch.randn(m, c, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    w1, b1 = torch.randn(hidden, c, generator=g) / 16, 0.1 * torch.randn(hidden, generator=g)
    w2, b2 = torch.randn(c, hidden, generator=g) / 32, 0.1 * torch.randn(c, generator=g)
    assert ops.tfm_ffn_fused_supported(c, hidden) and not ops.tfm_ffn_fused_supported(c, hidden + 64) and not ops.tfm_ffn_fused_supported(320, hidden)
    w1f, b1f = fold_layernorm(w1, b1, gamma, beta)
    p1, p2 = ops.PackedWeight(w1f, b1f), ops.PackedWeight(w2, b2)
    xd = x.to(DEV)
    f1, f2 = ops.tfm_pack_frag(p1), ops.tfm_pack_frag(p2)
    out = ops.tfm_ffn_fused(xd, p1, f1, p2, f2).cpu()
    again = ops.tfm_ffn_fused(xd, p1, f1, p2, f2).cpu()
    assert torch.equal(out, again)
    xd64 = x.double()
    n = F.layer_norm(xd64, (c,), gamma.double(), beta.double(), 1e-5)
    ref = (xd64 + F.gelu(n @ w1.double().T + b1.double()) @ w2.double().T + b2.double()).float()
    ident = (torch.ones(c, device=DEV), torch.zeros(c, device=DEV))
    n16 = ops.layernorm(xd, *ident, 1e-5, out_dtype=torch.float16)
    f16 = ops.linear(n16, p1, act="gelu", out_dtype=torch.float16)
    un = ops.linear(f16, p2, residual=xd).cpu()
    scale = float((ref - x).abs().max())          # the feed-forward term, without the residual that passes through in fp32
    e_ref, e_un = float((out - ref).abs().max()) / scale, float((out - un).abs().max()) / scale
    assert e_ref < 3e-3 and e_un < 2e-3, (e_ref, e_un)
    assert bool(torch.isfinite(out).all())


@pytest.mark.parametrize("m,k0", [(5504, 512), (45, 512), (1, 256), (8192, 512)])
def test_tfm_ffn_fused_with_output_projection(m, k0):
    """The same launch with the attention's output projection + residual as its prologue (x' = x + attn Wo^T + bo, never written
    to memory) against the fp64 definition and against linear(residual) followed by the plain fused launch."""
    import torch.nn.functional as F

    from astts import ops
    from astts.synth.model import fold_layernorm

    c, hidden = 256, 1024
    g = torch.Generator().manual_seed(m + k0)
    x = torch.randn(m, c, generator=g) * 2 + 0.3
    attn = torch.randn(m, k0, generator=g)
    wo, bo = torch.randn(c, k0, generator=g) / 24, 0.1 * torch.randn(c, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    w1, b1 = torch.randn(hidden, c, generator=g) / 16, 0.1 * torch.randn(hidden, generator=g)
    w2, b2 = torch.randn(c, hidden, generator=g) / 32, 0.1 * torch.randn(c, generator=g)
    w1f, b1f = fold_layernorm(w1, b1, gamma, beta)
    p1, p2, po = ops.PackedWeight(w1f, b1f), ops.PackedWeight(w2, b2), ops.PackedWeight(wo, bo)
    f1, f2, fo = ops.tfm_pack_frag(p1), ops.tfm_pack_frag(p2), ops.tfm_pack_frag(po)
    xd, ad = x.to(DEV), attn.to(DEV, torch.float16)
    out = ops.tfm_ffn_fused(xd, p1, f1, p2, f2, attn=ad, wo=po, wo_frag=fo).cpu()
    assert torch.equal(out, ops.tfm_ffn_fused(xd, p1, f1, p2, f2, attn=ad, wo=po, wo_frag=fo).cpu())
    x1 = x.double() + ad.cpu().double() @ wo.double().T + bo.double()
    n = F.layer_norm(x1, (c,), gamma.double(), beta.double(), 1e-5)
    ref = (x1 + F.gelu(n @ w1.double().T + b1.double()) @ w2.double().T + b2.double()).float()
    two = ops.tfm_ffn_fused(ops.linear(ad, po, residual=xd), p1, f1, p2, f2).cpu()
    scale = float((ref - x).abs().max())
    e_ref, e_two = float((out - ref).abs().max()) / scale, float((out - two).abs().max()) / scale
    assert e_ref < 3e-3 and e_two < 1e-3, (e_ref, e_two)
    assert bool(torch.isfinite(out).all())


@pytest.mark.parametrize("c,l,taps,dil", [(128, 700, 3, 1), (128, 700, 7, 3), (128, 515, 11, 5), (128, 40, 11, 5), (256, 300, 3, 1),
                                          (256, 300, 7, 5), (256, 129, 11, 3), (128, 1, 7, 1)])
def test_conv1d_snake_matches_definition(c, l, taps, dil):
    """astts_op_conv1d_snake (LDS-staged Snake + Conv1d + bias + residual + resblock-mean accumulation of the HiFT resblocks) against
    the fp64 definition: torch conv1d on the Snake-activated input, zero padding, dilation; fp32 and fp16 inputs / outputs; tiles
    with ragged tails, sequences shorter than the halo, the accumulate form."""
    import torch.nn.functional as F

    from astts import ops

    g = torch.Generator().manual_seed(c + l + taps + dil)
    b = 2
    x = torch.randn(b, l, c, generator=g)
    alpha = torch.rand(c, generator=g) * 2 + 0.1
    w = torch.randn(c, c, taps, generator=g) / math.sqrt(c * taps)
    bias = 0.1 * torch.randn(c, generator=g)
    res = torch.randn(b, l, c, generator=g)
    assert ops.conv1d_snake_supported(c, taps, dil) and not ops.conv1d_snake_supported(192, taps, dil) and not ops.conv1d_snake_supported(c, 4, 1)
    pw = ops.PackedWeight.from_conv1d(w, bias)
    wf = ops.conv_pack_frag(pw)
    xs = x.double() + torch.sin(alpha.double() * x.double()) ** 2 / (alpha.double() + 1e-9)
    ref = F.conv1d(xs.transpose(1, 2), w.double(), bias.double(), dilation=dil, padding=dil * (taps - 1) // 2).transpose(1, 2)
    scale = float(ref.abs().max())
    xd, rd = x.to(DEV), res.to(DEV)
    # snake + conv + bias + residual, fp32 in / out
    y = ops.conv1d_snake(xd, pw, wf, dil=dil, alpha=alpha.to(DEV), residual=rd).cpu()
    e = float((y - (ref + res.double()).float()).abs().max()) / scale
    assert e < 3e-3, e
    # fp16 input, fp16 output, no activation; accumulate form on top of an existing tensor
    x16 = xd.half()
    ref2 = F.conv1d(x16.cpu().double().transpose(1, 2), w.double(), bias.double(), dilation=dil, padding=dil * (taps - 1) // 2).transpose(1, 2)
    acc0 = torch.randn(b, l, c, generator=g)
    acc = acc0.to(DEV)
    y16 = ops.conv1d_snake(x16, pw, wf, dil=dil, out_dtype=torch.float16, acc=acc, acc_scale=1.0 / 3.0, acc_add=True)
    e2 = float((y16.float().cpu() - ref2.float()).abs().max()) / float(ref2.abs().max())
    e3 = float((acc.cpu() - (acc0.double() + ref2 / 3.0).float()).abs().max()) / float(ref2.abs().max())
    assert e2 < 3e-3 and e3 < 3e-3, (e2, e3)
    # accumulator only (no y), initialising form
    acc2 = torch.full((b, l, c), 7.0, device=DEV)
    assert ops.conv1d_snake(x16, pw, wf, dil=dil, want_y=False, acc=acc2, acc_scale=0.5) is None
    assert float((acc2.cpu() - (0.5 * ref2).float()).abs().max()) / float(ref2.abs().max()) < 3e-3


@pytest.mark.parametrize("b,t,ragged,cin", [(2, 70, True, 256), (16, 344, False, 256), (3, 33, True, 256), (1, 1, False, 256), (2, 688, True, 256),
                                             (2, 70, True, 512), (16, 344, False, 512), (48, 344, True, 256)])
def test_resnet_conv_block_matches_definition_and_five_launch_path(b, t, ragged, cin):
    """A ResnetBlock1D as three astts_op_resnet_conv launches (GroupNorm statistics taken by the producing convolution's epilogue,
    normalise + Mish + time-embedding add + mask applied by the consumer) against the fp64 definition (oracle semantics: statistics
    over the valid frames of each sequence) and against the five launches it replaces (conv, groupnorm, conv, groupnorm, conv)."""
    import torch.nn.functional as F

    from astts import ops

    c, groups = 256, 8
    g = torch.Generator().manual_seed(b * 100 + t)
    lens = torch.tensor([t] + [max(1, t - 9 * (i + 1)) for i in range(b - 1)]) if ragged else torch.full((b,), t)
    m = (torch.arange(t)[None, :] < lens[:, None]).float()[..., None]
    x = torch.randn(b, t, cin, generator=g) * m                      # cin = 512: the up blocks' [x | skip] concat
    w1, w2, wr = (torch.randn(c, ci, k, generator=g) / math.sqrt(ci * k) for ci, k in ((cin, 3), (c, 3), (cin, 1)))
    b1, b2_, br = (0.1 * torch.randn(c, generator=g) for _ in range(3))
    g1, be1, g2, be2 = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g), 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    tproj = torch.randn(b, c, generator=g)

    def block(h, w, bias, gam, bet):     # conv3 -> GroupNorm over the valid frames -> Mish, masked
        h = F.conv1d(h.transpose(1, 2), w.double(), bias.double(), padding=w.shape[-1] // 2).transpose(1, 2)
        out = torch.zeros_like(h)
        for i in range(b):
            L = int(lens[i])
            out[i, :L] = F.mish(F.group_norm(h[i:i + 1, :L].transpose(1, 2), groups, gam.double(), bet.double(), 1e-5)).transpose(1, 2)[0]
        return out

    xd = x.double()
    h = block(xd, w1, b1, g1, be1) + tproj.double()[:, None, :] * m.double()
    h = block(h, w2, b2_, g2, be2)
    ref = (F.conv1d(xd.transpose(1, 2), wr.double(), br.double()).transpose(1, 2) + h).float()

    dev = torch.device(DEV)
    p1, p2, pr = ops.PackedWeight.from_conv1d(w1, b1), ops.PackedWeight.from_conv1d(w2, b2_), ops.PackedWeight.from_conv1d(wr, br)
    f1, f2, fr = ops.conv_pack_frag(p1), ops.conv_pack_frag(p2), ops.conv_pack_frag(pr)
    xg, lg = x.to(dev), lens.to(dev, torch.int32)
    gd = [v.to(dev) for v in (g1, be1, g2, be2)]
    h1, s1 = ops.resnet_conv(xg, p1, f1, lens=lg, want_stats=True)
    h2, s2 = ops.resnet_conv(h1, p2, f2, lens=lg, in_gn=(s1, gd[0], gd[1]), in_add=tproj.to(dev), want_stats=True)
    out = ops.resnet_conv(xg, pr, fr, lens=lg, res_gn=(h2, s2, gd[2], gd[3])).cpu()
    # the five launches it replaces
    o1 = ops.conv1d(xg, p1, pad=1)
    o1 = ops.groupnorm(o1, gd[0], gd[1], groups, 1e-5, lens=lg, mish=True, add_bc=tproj.to(dev), out_dtype=torch.float16)
    o2 = ops.conv1d(o1, p2, pad=1)
    o2 = ops.groupnorm(o2, gd[2], gd[3], groups, 1e-5, lens=lg, mish=True)
    five = ops.conv1d(xg, pr, residual=o2).cpu()
    scale = float(ref.abs().max())
    for i in range(b):
        L = int(lens[i])
        e_ref = float((out[i, :L] - ref[i, :L]).abs().max()) / scale
        e_five = float((out[i, :L] - five[i, :L]).abs().max()) / scale
        assert e_ref < 4e-3 and e_five < 4e-3, (i, e_ref, e_five)
    assert torch.equal(out, ops.resnet_conv(xg, pr, fr, lens=lg, res_gn=(h2, s2, gd[2], gd[3])).cpu())
    assert bool(torch.isfinite(out).all())
    if b * ((t + 31) // 32) >= 512:
        # a grid of two rounds of workgroups takes the two-workgroups-per-CU form (weights in half units): the same operations in the same
        # order, so a sequence's rows equal, bit for bit, what the one-workgroup-per-CU form computes for it in a batch of two
        h1s, s1s = ops.resnet_conv(xg[:2], p1, f1, lens=lg[:2], want_stats=True)
        h2s, s2s = ops.resnet_conv(h1s, p2, f2, lens=lg[:2], in_gn=(s1s, gd[0], gd[1]), in_add=tproj[:2].to(dev), want_stats=True)
        outs = ops.resnet_conv(xg[:2], pr, fr, lens=lg[:2], res_gn=(h2s, s2s, gd[2], gd[3])).cpu()
        assert torch.equal(h1[:2].cpu(), h1s.cpu()) and torch.equal(h2[:2].cpu(), h2s.cpu()) and torch.equal(out[:2], outs)


def test_cu_masked_stream_runs_kernels_with_the_same_results():
    """astts_stream_create_cu_mask (CU partitions between concurrent stages: scripts/cu_mask_probe.py): a GEMM enqueued on a
    stream restricted to 64 CUs gives the bits of the unrestricted launch, and a bad mask is refused."""
    import ctypes

    from astts import _lib, ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(300, 256, generator=g).half().to(DEV)
    pw = ops.PackedWeight(torch.randn(512, 256, generator=g) / 16, torch.randn(512, generator=g))
    ref = ops.linear(x, pw)
    torch.cuda.synchronize()
    st = ops.cu_masked_stream(64)
    with torch.cuda.stream(st):
        out = ops.linear(x, pw)
    st.synchronize()
    assert torch.equal(out, ref)
    h = ctypes.c_void_p()
    assert _lib.load().astts_stream_create_cu_mask(None, 8, ctypes.byref(h)) == _lib.ERR_INVALID


@pytest.mark.gpu
def test_lane_exchanges_equal_the_shuffle_form():
    """csrc/xlane.h: the decode-step reductions exchange lanes with DPP modifiers and v_permlane16/32_swap instead of
    ds_bpermute.  The library's self-test compares every butterfly offset and the composed sums / maxima with __shfl_xor on
    pseudo-random values: bit for bit (the swaps are inline assembly because the compiler folded the builtin's two results
    into one -- this is the guard for that)."""
    from astts import ops

    assert ops.selftest_xlane() == 0
