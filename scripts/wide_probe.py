"""The wide decode engine (33..256 rows, plain GEMMs) alone: ms per step at WP_ROWS rows and a WP_PREFIX-position prefix; under
rocprofv3 --kernel-trace --stats the per-kernel sums say where a step's time sits."""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_lm_weights
from astts.synth.model import AcousticLM
cfg = SynthConfig()
lm = AcousticLM(make_lm_weights(cfg, 0), cfg, torch.device('cuda'))
B, S0, N = int(os.environ.get('WP_ROWS', 128)), int(os.environ.get('WP_PREFIX', 185)), int(os.environ.get('WP_STEPS', 66))
g = torch.Generator(device='cuda').manual_seed(0)
pre = torch.randn(S0, B, cfg.lm_dim, device='cuda', generator=g)
u = torch.rand(N, B, 2, device='cuda', generator=g)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lm.decode(pre, N, u, True, wide=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
lm.decode(pre, 2, u[:2], True, wide=True)
torch.cuda.synchronize(); d2 = time.perf_counter() - t0
print(f'wide engine: B={B} prefix {S0}: per step {(dt - d2) / (N - 2) * 1e3:.3f} ms ({(dt - d2) / (N - 2) / B * 32 * 1e3:.3f} ms per 32 rows)', flush=True)
