import sys, time, math, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_all
from astts.synth.model import SynthEngine, PipelinedSynth
cfg = SynthConfig(); W = make_all(cfg, 0); eng = SynthEngine(W, cfg, 'cuda'); del W
g = torch.Generator(device='cuda').manual_seed(0)
B, Tt, Tp, Ts = 8, 32, 150, 250
dev = 'cuda'
text = torch.randint(0, cfg.text_vocab, (B, Tt), device=dev, generator=g); tlen = torch.full((B,), Tt, dtype=torch.int32, device=dev)
spk_s = torch.randn(B, cfg.spk_dim, device=dev, generator=g); spk_t = torch.randn(B, cfg.spk_dim, device=dev, generator=g)
style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g); timbre_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g)
tmp = cfg.mel_frames_for_tokens(Tp); tm = cfg.mel_frames_for_tokens(Ts)
timbre_mel = torch.randn(B, tmp, cfg.mel, device=dev, generator=g)
u = torch.rand(Ts, B, 2, device=dev, generator=g); z = torch.randn(B, tmp + tm, cfg.mel, device=dev, generator=g)
nh = cfg.nb_harmonics + 1
phase0 = (torch.rand(B, nh, device=dev, generator=g) * 2 - 1) * math.pi; phase0[:, 0] = 0
noise = torch.randn(B, tm * cfg.upsample_total, nh, device=dev, generator=g)
args = (text, tlen, spk_s, style_tok, Ts, u, timbre_tok, timbre_mel, spk_t, z, phase0, noise)
for _ in range(2): ref = eng.tts(*args)
torch.cuda.synchronize()
K = 6
t0 = time.perf_counter()
for _ in range(K): out = eng.tts(*args)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
audio = B * ref[2].shape[1] / cfg.sample_rate
print(f'sequential: {dt / K * 1e3:.1f} ms/step RTF^-1 {audio * K / dt:.1f}')
def timed(pipe, tag, K=12):
    t0 = time.perf_counter(); outs = []
    for _ in range(K):
        r = pipe.submit(*args)
        if r is not None: outs.append(r)
    outs += pipe.drain()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{tag}: {dt / K * 1e3:.1f} ms/step RTF^-1 {audio * K / dt:.1f}', [int(s.cuda_stream) % 100000 for s in pipe.s_lm + [pipe.s_render]], flush=True)

from astts import ops
def timed(pipe, tag, K=16):
    with torch.cuda.stream(pipe.front_stream):
        for _ in range(3): pipe.submit(*args)
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter(); outs = []
        for _ in range(K):
            r = pipe.submit(*args)
            if r is not None: outs.append(r)
        outs += pipe.drain()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{tag}: {dt / K * 1e3:.1f} ms/step RTF^-1 {audio * K / dt:.1f}', flush=True)
for trial in range(2):
    for depth, rd in ((2, 1), (2, 2), (3, 2), (2, 3), (4, 2)):
        timed(PipelinedSynth(eng, lm_depth=depth, lm_priority=0, render_priority=0, render_depth=rd), f'LM chains {depth}, render streams {rd}')
