"""Where does the pipelined step go?  render-only, LM-only (1..3 concurrent chains), and both."""
import sys, time, math, threading
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_all
from astts.synth.model import SynthEngine
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
toks = eng.tts_tokens(text, tlen, spk_s, style_tok, Ts, u)
torch.cuda.synchronize()

def lm_loop(n, stream):
    with torch.cuda.stream(stream):
        for _ in range(n):
            eng.tts_tokens(text, tlen, spk_s, style_tok, Ts, u)
        stream.synchronize()

def render_loop(n, stream, host=None):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        for _ in range(n):
            eng.tts_render(toks, timbre_tok, timbre_mel, spk_t, z, phase0, noise)
        if host is not None: host.append((time.perf_counter() - t0) / n)
        stream.synchronize()

from astts import ops

def run(tag, lm_streams, render_streams, K=4):
    host = []
    th = [threading.Thread(target=lm_loop, args=(K, st)) for st in lm_streams]
    th += [threading.Thread(target=render_loop, args=(K, st, host)) for st in render_streams]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{tag}: {len(lm_streams)} LM x {K} + {len(render_streams)} render x {K}: {dt*1e3/K:.1f} ms per round', flush=True)

def plain(n, prio=0): return [torch.cuda.Stream(priority=prio) for _ in range(n)]
import os
n_lm = int(os.environ.get('N_LM', '3')); prio = int(os.environ.get('LM_PRIO', '-1'))
order = os.environ.get('ORDER', 'lm_first')
if order == 'lm_first':
    lm = plain(n_lm, prio); rd = plain(1)
else:
    rd = plain(1); lm = plain(n_lm, prio)
run(f'n_lm={n_lm} prio={prio} {order}', lm, [])
run(f'n_lm={n_lm} prio={prio} {order}', lm, rd)
run(f'n_lm={n_lm} prio={prio} {order}', lm, rd, K=8)
