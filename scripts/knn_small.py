import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.knn import StyleBank
g = torch.Generator(device='cuda').manual_seed(1234)
n, d, q, k = int(os.environ.get('KNN_N', 1000)), int(os.environ.get('KNN_D', 6144)), int(os.environ.get('KNN_Q', 8)), 3
bank = torch.randn((n, d), generator=g, device='cuda').to(torch.float16)
sb = StyleBank(bank)
qs = bank[torch.randint(0, n, (q,), generator=g, device='cuda')].float() + 0.5 * torch.randn((q, d), generator=g, device='cuda')
oi = torch.empty((q, k), dtype=torch.int64, device='cuda'); os_ = torch.empty((q, k), dtype=torch.float32, device='cuda')
for _ in range(int(os.environ.get('KNN_ITERS', 200))): sb.search_device(qs, k, out_idx=oi, out_score=os_)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = int(os.environ.get('KNN_ITERS', 2000))
for _ in range(N): sb.search_device(qs, k, out_idx=oi, out_score=os_)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / N
print(f'search (N={n}, D={d}, Q={q}, k={k}): {us:.2f} us per search = {q / us * 1e6:.0f} QPS')
