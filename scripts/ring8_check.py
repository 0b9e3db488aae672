"""gemm_ring8 (ring mode 5; the launcher's 256 x 256 tile) against gemm_ring<4,2,2,4,8> (forced mode 4): the same tile and the same accumulation order per accumulator, so
the outputs must be BIT-IDENTICAL; repeated launches on changing inputs look for a missed LDS-DMA wait (a race shows as a flicker)."""
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops

bad = 0
for m, k, n in [(256, 64, 256), (256, 128, 256), (300, 192, 260), (5000, 1024, 4097), (15360, 3072, 5120), (4096, 64, 4096), (8192, 6144, 2048), (70000, 256, 1536)]:
    pw = ops.PackedWeight(torch.randn(n, k) / 8, torch.randn(n) * 0.1)
    for rep in range(12):
        x = torch.randn(m, k, device='cuda').half()
        ops.set_gemm_ring_mode(4)
        y4 = ops.gemm(x, pw, out=torch.empty((m, n), dtype=torch.float32, device='cuda')).clone()
        ops.set_gemm_ring_mode(5)
        y5 = ops.gemm(x, pw, out=torch.empty((m, n), dtype=torch.float32, device='cuda'))
        torch.cuda.synchronize()
        if not torch.equal(y4, y5):
            d = (y4 - y5).abs()
            bad += 1
            idx = (d > 0).nonzero()
            print(f"MISMATCH m={m} k={k} n={n} rep {rep}: {idx.shape[0]} elements, max |d| {float(d.max()):.3e}; first {idx[:4].tolist()} rows mod 256 {sorted(set((idx[:, 0] % 256 // 32).tolist()))} cols mod 256 {sorted(set((idx[:, 1] % 256 // 32).tolist()))}")
            break
    else:
        print(f"m={m} k={k} n={n}: 12 x bit-identical to mode 4")
ops.set_gemm_ring_mode(-1)
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
