"""Full-size vocoder + flow.mu only (for rocprofv3 traces)."""
import sys, time, os, math
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_flow_weights, make_hift_weights
from astts.synth.model import FlowDecoder, HiftVocoder
cfg = SynthConfig()
dev = torch.device('cuda')
hv = HiftVocoder(make_hift_weights(cfg, 0), cfg, dev)
fd = FlowDecoder(make_flow_weights(cfg, 0), cfg, dev)
g = torch.Generator(device='cuda').manual_seed(0)
B, tm = 8, 430
mel = torch.randn(B, tm, cfg.mel, device=dev, generator=g)
nh = cfg.nb_harmonics + 1
phase0 = (torch.rand(B, nh, device=dev, generator=g) * 2 - 1) * math.pi; phase0[:, 0] = 0
noise = torch.randn(B, tm * cfg.upsample_total, nh, device=dev, generator=g)
tok = torch.randint(0, cfg.speech_vocab, (B, 400), device=dev, generator=g).to(torch.int32); tl = torch.full((B,), 400, dtype=torch.int32, device=dev)
for _ in range(2):
    hv.forward(mel, phase0, noise); fd.mu(tok, tl, 688)
torch.cuda.synchronize()
print('MARK')
t0 = time.perf_counter(); hv.forward(mel, phase0, noise); torch.cuda.synchronize(); t1 = time.perf_counter()
fd.mu(tok, tl, 688); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'hift {1e3*(t1-t0):.2f} ms, mu {1e3*(t2-t1):.2f} ms')
