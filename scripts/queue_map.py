import sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops, _lib
lib = _lib.load()
dev = torch.device('cuda')
torch.zeros(1, device=dev)
ss = [torch.cuda.Stream() for _ in range(20)]
default = torch.cuda.current_stream()
def t_pair(a, b, us=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.astts_stream_spin(us, int(a.cuda_stream)); lib.astts_stream_spin(us, int(b.cuda_stream))
    a.synchronize(); b.synchronize()
    return (time.perf_counter() - t0) * 1e6
lib.astts_stream_spin(10, int(ss[0].cuda_stream)); torch.cuda.synchronize()
print('single', [round(t_pair(s, s)) for s in ss[:4]])
allst = [default] + ss
for i, a in enumerate(allst[:13]):
    print(f'{i:2d}', ' '.join('S' if t_pair(a, b) > 480 else '.' for b in allst[:13]))
