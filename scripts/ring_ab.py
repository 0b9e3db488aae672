"""A/B of ring GEMM tiles in ONE process: the modes of RING_MODES alternate (m0, m1, m0, m1, ...) for ROUNDS rounds of 20 launches on
each shape, median per mode -- the first kernel measured on a cold chip reads ~15 % low, so ring_shapes.py's one pass per mode cannot
rank two tiles that are within that of each other."""
import os
import statistics
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops

modes = [int(v) for v in os.environ.get('RING_MODES', '4,5').split(',')]
rounds = int(os.environ.get('ROUNDS', '7'))
uniform = os.environ.get('UNIFORM', '0') == '1'
if os.environ.get('RING_SHAPES'):          # "m,k,n;m,k,n;..."
    shapes_env = [tuple(int(v) for v in t.split(',')) for t in os.environ['RING_SHAPES'].split(';')]
shapes = shapes_env if os.environ.get('RING_SHAPES') else [(4096, 4096, 4096), (8192, 8192, 8192), (15360, 3072, 5120), (15360, 8192, 3072), (15360, 3072, 3072), (23680, 1024, 4096), (23680, 4096, 1024),
          (1920, 3072, 5120), (44032, 256, 1536), (256, 6144, 100000)]


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for m, k, n in shapes:
    x = (torch.rand(m, k, device='cuda') * 2 - 1 if uniform else torch.randn(m, k, device='cuda')).half()
    pw = ops.PackedWeight(torch.rand(n, k) * 2 - 1 if uniform else torch.randn(n, k) / 32, torch.randn(n) * 0.1)
    out = torch.empty((m, n), dtype=torch.float16, device='cuda')
    ts = {mode: [] for mode in modes}
    for _ in range(rounds):
        for mode in modes:
            ops.set_gemm_ring_mode(mode)
            ts[mode].append(timed(lambda: ops.gemm(x, pw, out=out)))
    ops.set_gemm_ring_mode(-1)
    print(f"M={m:6d} K={k:5d} N={n:6d}  " + "   ".join(
        f"{mode}: {statistics.median(t):8.1f} us {2.0 * m * n * k / statistics.median(t) * 1e-6:6.0f} TF (best {2.0 * m * n * k / min(t) * 1e-6:5.0f})" for mode, t in ts.items()), flush=True)
