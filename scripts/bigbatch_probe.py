"""Is a decode step over 256 rows at once (plain GEMMs + per-row attention, AcousticLM.forward_new: the operator path) cheaper on the GPU
than eight 32-row engine chains?  Times both for a few steps; under rocprofv3 --kernel-trace --stats the kernel sums give the GPU-side cost
of the operator path without its Python host loop."""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_lm_weights
from astts.synth.model import AcousticLM
cfg = SynthConfig()
lm = AcousticLM(make_lm_weights(cfg, 0), cfg, torch.device('cuda'))
B, S0, N = int(os.environ.get('BB_ROWS', 256)), 185, int(os.environ.get('BB_STEPS', 16))
g = torch.Generator(device='cuda').manual_seed(0)
pre = torch.randn(S0, B, cfg.lm_dim, device='cuda', generator=g)
u = torch.rand(N, B, 2, device='cuda', generator=g)
for mode in os.environ.get('BB_MODES', 'ops,engine').split(','):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks = lm.decode(pre, N, u, True, use_engine=(mode == 'engine'))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{mode}: B={B} prefill + {N} steps {dt * 1e3:.1f} ms', flush=True)
    # steps only: difference between N and 2 steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lm.decode(pre, 2, u[:2], True, use_engine=(mode == 'engine'))
    torch.cuda.synchronize(); d2 = time.perf_counter() - t0
    print(f'{mode}: per step {(dt - d2) / (N - 2) * 1e3:.3f} ms ({(dt - d2) / (N - 2) / B * 32 * 1e3:.3f} ms per 32 rows)', flush=True)
