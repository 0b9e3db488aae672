"""Full-size flow solve only (for rocprofv3 traces): 1 warm-up + N timed solves."""
import sys, time, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_flow_weights
from astts.synth.model import FlowDecoder
cfg = SynthConfig()
fd = FlowDecoder(make_flow_weights(cfg, 0), cfg, torch.device('cuda'))
g = torch.Generator(device='cuda').manual_seed(0)
B, T = int(os.environ.get('FLOW_B', '8')), int(os.environ.get('FLOW_T', '688'))
dev = 'cuda'
z = torch.randn(B, T, cfg.mel, device=dev, generator=g); mu = torch.randn(B, T, cfg.mel, device=dev, generator=g)
cond = torch.randn(B, T, cfg.mel, device=dev, generator=g); spk = torch.randn(B, cfg.mel, device=dev, generator=g)
fd.solve(z.clone(), mu, spk, cond); torch.cuda.synchronize()
N = int(os.environ.get('FLOW_N', '2'))
t0 = time.perf_counter()
for _ in range(N): fd.solve(z.clone(), mu, spk, cond)
torch.cuda.synchronize()
print(f'flow solve B={B} T={T}: {(time.perf_counter() - t0) / N * 1e3:.2f} ms')
