"""gemm_ring vs reference on the flow decoder's projection shapes (+ odd shapes), and rough timing."""
import sys, os, math, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
import torch.nn.functional as F
from astts import ops
dev = 'cuda'
shapes = [(5504, 256, 1536, 'none', True, False), (5504, 256, 1024, 'gelu', True, False), (5504, 512, 256, 'none', False, True),
          (5504, 1024, 256, 'none', False, True), (11008, 256, 1536, 'none', True, False), (11008, 1024, 256, 'none', False, True),
          (100, 64, 80, 'none', False, False), (64, 128, 33, 'silu', True, False), (129, 192, 100, 'none', False, True),
          (1000, 320, 512, 'relu', False, False), (777, 1024, 4097, 'none', True, False), (5520, 256, 512, 'none', False, False)]
print('ASTTS_GEMM_RING =', os.environ.get('ASTTS_GEMM_RING'))
for m, k, n, act, o16, res in shapes:
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g).half(); w = torch.randn(n, k, generator=g) / math.sqrt(k); b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g) if res else None
    pw = ops.PackedWeight(w, b)
    xd = x.to(dev); rd = None if r is None else r.to(dev)
    y = ops.linear(xd, pw, act=act, residual=rd, out_dtype=torch.float16 if o16 else torch.float32)
    ref = F.linear(x.float(), w.half().float(), b)
    ref = {'none': lambda t: t, 'gelu': F.gelu, 'silu': F.silu, 'relu': F.relu}[act](ref)
    if res: ref = ref + r
    err = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
    tol = 2e-3 if o16 else 2e-4
    torch.cuda.synchronize()
    n_it = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.empty_like(y)
    e0.record()
    for _ in range(n_it): ops.gemm(xd, pw, act=act, residual=rd, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n_it
    print(f'm={m:6d} k={k:5d} n={n:5d} act={act:5s} out16={int(o16)} res={int(res)}: rel err {err:.2e} {"OK" if err < tol else "FAIL"}   {us:7.2f} us  {2.0*m*n*k/us/1e6:7.1f} TFLOP/s', flush=True)
