"""kNN throughput at the BASELINE config shapes (bank resident, queries on device)."""
import sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.knn import StyleBank
def run(n, d, q, k=3, reps=20):
    g = torch.Generator(device='cuda').manual_seed(1234)
    bank = torch.randn((n, d), generator=g, device='cuda').to(torch.float16)
    sb = StyleBank(bank)
    qs = bank[torch.randint(0, n, (q,), generator=g, device='cuda')].float() + 0.5 * torch.randn((q, d), generator=g, device='cuda')
    oi = torch.empty((q, k), dtype=torch.int64, device='cuda'); os_ = torch.empty((q, k), dtype=torch.float32, device='cuda')
    for _ in range(3): sb.search_device(qs, k, out_idx=oi, out_score=os_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): sb.search_device(qs, k, out_idx=oi, out_score=os_)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    sb.profile_enable(True)
    for _ in range(reps): sb.search_device(qs, k, out_idx=oi, out_score=os_)
    torch.cuda.synchronize(); ms, cnt = sb.profile_read(); sb.profile_enable(False)
    scan_us = ms * 1e3 / max(cnt, 1)
    byts = n * d * 2 + q * d * 4 + q * k * 12
    groups = (q + 255) // 256
    print(f'N={n:7d} D={d:5d} Q={q:4d}: {dt * 1e6:9.1f} us/search  {q / dt:10.0f} QPS | scan {scan_us:8.1f} us/launch x{cnt // reps}  '
          f'{byts / groups / (scan_us * 1e-6) / 1e9:7.0f} GB/s algorithmic, {2.0 * min(q, 256) * n * d / (scan_us * 1e-6) / 1e12:6.1f} TFLOP/s | fallbacks {sb.last_fallbacks()}')
for shp in [(1000, 6144, 8), (1000, 6144, 64), (100000, 768, 256), (100000, 6144, 8), (100000, 6144, 32), (100000, 6144, 256)]:
    run(*shp)
