"""Counter target for the 256 x 256 ring GEMMs: RING_PMC_SHAPE (default 8192^3) in ring modes 4 (one barrier per K tile) and 5 (the
eight-phase schedule), four launches each after two of warm-up; run under rocprofv3 --kernel-trace --pmc <group>."""
import os
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops

m, k, n = [int(v) for v in os.environ.get('RING_PMC_SHAPE', '8192,8192,8192').split(',')]
x = torch.randn(m, k, device='cuda').half()
pw = ops.PackedWeight(torch.randn(n, k) / 32, torch.randn(n) * 0.1)
out = torch.empty((m, n), dtype=torch.float16, device='cuda')
for mode in [int(v) for v in os.environ.get('RING_MODES', '4,5').split(',')]:
    ops.set_gemm_ring_mode(mode)
    for _ in range(6):
        ops.gemm(x, pw, out=out)
    torch.cuda.synchronize()
ops.set_gemm_ring_mode(-1)
