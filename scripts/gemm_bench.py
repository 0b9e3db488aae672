"""Time astts_op_gemm on the shapes of the flow estimator / vocoder / LM (HIP events, 50 reps each)."""
import sys, math
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops

def bench(m, k, n, taps=1, reps=50, t=None):
    x = torch.randn(m, k, device='cuda')
    w = ops.PackedWeight(torch.randn(n, taps, k) / math.sqrt(k * taps), torch.randn(n))
    kw = {}
    if taps > 1:
        kw = dict(t_in=t, t_out=t, pad=(taps - 1) // 2)
    y = ops.gemm(x, w, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(x, w, out=y, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * m * n * k * taps
    print(f'M={m:6d} K={k:5d} N={n:5d} taps={taps}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s')

shapes = [(5504, 256, 1536), (5504, 512, 256), (5504, 256, 1024), (5504, 1024, 256), (11008, 256, 1536), (11008, 1024, 256),
          (11008, 256, 1024), (1480, 1024, 3072), (1480, 1024, 4096), (3200, 512, 2048), (8, 1024, 3072), (8, 4096, 1024), (8, 1024, 4096)]
for s in shapes:
    bench(*s)
bench(5504, 256, 256, taps=3, t=344)
bench(11008, 320, 256, taps=3, t=688)
bench(27520, 256, 256, taps=7, t=3440)
bench(220160, 128, 128, taps=11, t=27520)
bench(220160, 128, 128, taps=3, t=27520)
