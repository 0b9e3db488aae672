import sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.knn import StyleBank
g = torch.Generator(device='cuda').manual_seed(1234)
n, d, q, k = 1000, 6144, 8, 3
bank = torch.randn((n, d), generator=g, device='cuda').to(torch.float16)
sb = StyleBank(bank)
qs = bank[torch.randint(0, n, (q,), generator=g, device='cuda')].float() + 0.5 * torch.randn((q, d), generator=g, device='cuda')
oi = torch.empty((q, k), dtype=torch.int64, device='cuda'); os_ = torch.empty((q, k), dtype=torch.float32, device='cuda')
for _ in range(200): sb.search_device(qs, k, out_idx=oi, out_score=os_)
torch.cuda.synchronize()
