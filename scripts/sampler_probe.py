"""Per-launch time of astts_op_ras_sample at the decode step's shape (8 rows x 4097 logits, history of 100 tokens)."""
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
b, v = 8, 4097
g = torch.Generator(device=dev).manual_seed(0)
logits = torch.randn(b, v, device=dev, generator=g) * 2
hist = torch.randint(0, 4096, (b, 512), device=dev, dtype=torch.int32, generator=g)
u = torch.rand(b, 2, device=dev, generator=g)
out = torch.empty(b, dtype=torch.int32, device=dev)


def timed(fn, n=500):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for hl, win in ((100, 10), (100, 0), (0, 10)):
    t = timed(lambda: ops.ras_sample(logits, hist, hl, u, 25, 0.8, win, 0.1, 4096, True, out=out))
    print(f'hist_len={hl} win={win}: {t:.2f} us per launch (incl. boundary)')
