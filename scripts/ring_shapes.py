"""fp16-activation GEMMs on the LDS-DMA ring kernel, forced tile (1: 128x128, 2: 128x64, 3: 64x64, 4: 256x256 with eight waves and one barrier per K tile,
5: 256x256 on the eight-phase schedule) against the launcher's own choice (-1): us per launch and TFLOP/s on shapes of the embedder, the LM prefill, the
64-sequence flow and the kNN scan.  ONE pass per mode: the first kernel on a cold chip reads ~15 % low -- scripts/ring_ab.py alternates two modes."""
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
shapes = [(15360, 3072, 5120), (15360, 3072, 16384), (15360, 8192, 3072), (15360, 3072, 3072), (1920, 3072, 5120), (23680, 1024, 3072), (23680, 1024, 4096),
          (23680, 4096, 1024), (22016, 256, 1536), (22016, 256, 1024), (22016, 1024, 256), (44032, 256, 1536), (5504, 256, 1536), (256, 6144, 100000)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for m, k, n in shapes:
    x = torch.randn(m, k, device='cuda').half()
    pw = ops.PackedWeight(torch.randn(n, k) / 32, torch.randn(n) * 0.1)
    out = torch.empty((m, n), dtype=torch.float16, device='cuda')
    row = []
    for mode in [int(v) for v in __import__('os').environ.get('RING_MODES', '-1,1,4,5').split(',')]:
        ops.set_gemm_ring_mode(mode)
        t = timed(lambda: ops.gemm(x, pw, out=out))
        row.append(f"{'auto' if mode < 0 else mode}: {t:8.1f} us {2.0 * m * n * k / t * 1e-6:6.0f} TF")
    ops.set_gemm_ring_mode(-1)
    print(f"M={m:6d} K={k:5d} N={n:6d}  " + "   ".join(row), flush=True)
