"""residual GEMM + LayerNorm as two launches vs the fused row-complete kernel (astts_op_gemm_ln), for a rocprofv3 kernel trace."""
import sys, math
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
for m, k in ((5504, 512), (5504, 1024), (11008, 1024)):
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).half().to(dev); w = ops.PackedWeight(torch.randn(256, k, generator=g) / math.sqrt(k), torch.randn(256, generator=g))
    r = torch.randn(m, 256, generator=g).to(dev); ga = torch.ones(256, device=dev); be = torch.zeros(256, device=dev)
    for _ in range(60):
        y = ops.linear(x, w, residual=r)
        n = ops.layernorm(y, ga, be, 1e-5, out_dtype=torch.float16)
    for _ in range(60):
        y2, n2 = ops.linear_ln(x, w, r, (ga, be))
    torch.cuda.synchronize()
    print(m, k, float((y - y2).abs().max()), float((n.float() - n2.float()).abs().max()))
