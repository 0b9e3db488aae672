"""Kernel-level time of the fused GroupNorm (+ Mish) at the flow estimator's shape (run under rocprofv3 --kernel-trace)."""
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
x = torch.randn(16, 344, 256, device=dev)
g, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
lens = torch.full((16,), 344, dtype=torch.int32, device=dev)
for eps in (1e-5, -1.0, -2.0):
    for _ in range(50):
        ops.groupnorm(x, g, b, 8, eps, lens=lens, mish=True, out_dtype=torch.float16)
    torch.cuda.synchronize()
