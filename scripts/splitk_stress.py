"""Stress the split-K skinny GEMM: many launches back to back (optionally under a concurrent load) against the
unsplit kernel; a lost / stale partial sum shows as an O(1) error."""
import sys, os, math, threading, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops, _lib
dev = 'cuda'
g = torch.Generator().manual_seed(0)
m, k, n = int(os.environ.get('M', '8')), 4096, 1024
w = torch.randn(n, k, generator=g) / math.sqrt(k); b = torch.randn(n, generator=g)
pw = ops.PackedWeight(w, b)
lib = _lib.load()
xs = [torch.randn(m, k, generator=g).to(dev) for _ in range(16)]
res = torch.randn(m, n, generator=g).to(dev)
def unsplit(x):
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    _lib.check(lib.astts_op_gemm_fused(x.data_ptr(), None, None, None, 0.0, pw.data.data_ptr(), pw.bias.data_ptr(), res.data_ptr(), out.data_ptr(), None, 0,
                                       m, n, 0, k, k, k, n, 0, n, 0, 1.0, 0.1, _lib.stream_ptr()))
    return out
refs = [unsplit(x) for x in xs]
torch.cuda.synchronize()
stop = False
def load():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a = torch.randn(4096, 4096, device=dev)
        while not stop:
            for _ in range(20): a = (a @ a) * 1e-3
            s.synchronize()
for with_load in (False, True):
    stop = False
    th = threading.Thread(target=load) if with_load else None
    if th: th.start(); time.sleep(0.2)
    bad = 0; worst = 0.0; N = int(os.environ.get('N', '20000'))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        outs = []
        t0 = time.perf_counter()
        for i in range(N):
            y = ops.gemm_fused(xs[i % 16], pw, m, residual=res)
            outs.append((i % 16, y))
            if len(outs) == 512:
                s.synchronize()
                for j, yy in outs:
                    d = float((yy - refs[j]).abs().max())
                    worst = max(worst, d)
                    bad += d > 1e-3
                outs = []
        s.synchronize(); dt = time.perf_counter() - t0
    stop = True
    if th: th.join()
    print(f'load={with_load}: {N} launches, {bad} bad, worst |d| {worst:.3g}, {dt / N * 1e6:.1f} us/launch', flush=True)
