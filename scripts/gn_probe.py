"""Per-launch time of the fused GroupNorm (+ Mish) at the flow estimator's shapes."""
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
for t in (344, 688):
    x = torch.randn(16, t, 256, device=dev)
    g, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    lens = torch.full((16,), t, dtype=torch.int32, device=dev)
    for dt in (torch.float32, torch.float16):
        fn = lambda: ops.groupnorm(x, g, b, 8, 1e-5, lens=lens, mish=True, out_dtype=dt)
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f'T={t} out={dt}: {e0.elapsed_time(e1) * 1e3 / 300:.2f} us per launch (incl. boundary and the output allocation)')
