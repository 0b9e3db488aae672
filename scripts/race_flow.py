"""Repro harness: flow solve on one stream while another host thread drives a second stream with some load."""
import sys, time, math, threading, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_all
from astts.synth.model import SynthEngine
from astts import ops
cfg = SynthConfig(); W = make_all(cfg, 0); eng = SynthEngine(W, cfg, 'cuda'); del W
g = torch.Generator(device='cuda').manual_seed(0)
B, Tt, Tp, Ts = int(os.environ.get("RACE_B", 8)), 32, 150, 250
dev = 'cuda'
text = torch.randint(0, cfg.text_vocab, (B, Tt), device=dev, generator=g); tlen = torch.full((B,), Tt, dtype=torch.int32, device=dev)
spk_s = torch.randn(B, cfg.spk_dim, device=dev, generator=g)
style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g)
u = torch.rand(Ts, B, 2, device=dev, generator=g)
T = 688
z = torch.randn(B, T, cfg.mel, device=dev, generator=g); mu = torch.randn(B, T, cfg.mel, device=dev, generator=g)
cond = torch.randn(B, T, cfg.mel, device=dev, generator=g); spk = torch.randn(B, cfg.mel, device=dev, generator=g)
fd = eng.flow
ref = fd.solve(z.clone(), mu, spk, cond); torch.cuda.synchronize()
ref2 = fd.solve_ops(z.clone(), mu, spk, cond); torch.cuda.synchronize()
print('engine vs ops (idle GPU):', float((ref - ref2).abs().max()))
stop = False
def load(kind, stream):
    with torch.cuda.stream(stream):
        a = torch.randn(4096, 4096, device=dev)
        while not stop:
            if kind == 'lm': eng.tts_tokens(text, tlen, spk_s, style_tok, Ts, u)
            elif kind == 'prefix': eng.lm.prefix(text, tlen, spk_s, style_tok)
            elif kind == 'flow_ops': fd.solve_ops(z.clone(), mu, spk, cond)
            elif kind == 'flow_eng': fd.solve(z.clone(), mu, spk, cond)
            elif kind == 'matmul':
                for _ in range(50): a = (a @ a) * 1e-3
            stream.synchronize()
for kind in os.environ.get('LOADS', 'none,lm,prefix,matmul,flow_ops,flow_eng').split(','):
    stop = False
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    th = threading.Thread(target=load, args=(kind, sB)); th.start()
    time.sleep(0.3)
    res = []
    with torch.cuda.stream(sA):
        for path in ('engine', 'ops'):
            for it in range(4):
                out = (fd.solve if path == 'engine' else fd.solve_ops)(z.clone(), mu, spk, cond)
                sA.synchronize()
                res.append((path, float((out - ref).abs().max())))
    stop = True; th.join(); torch.cuda.synchronize()
    print(f'load={kind}:', ' '.join(f'{p}:{d:.3g}' for p, d in res), flush=True)
