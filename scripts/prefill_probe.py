"""What the LM stage spends before its first decode step (prefix assembly, prefill, first logits), and host time of each."""
import sys, time, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_all
from astts.synth.model import SynthEngine
cfg = SynthConfig()
W = make_all(cfg, 0)
eng = SynthEngine(W, cfg, 'cuda'); torch.cuda.synchronize()
del W
g = torch.Generator(device='cuda').manual_seed(0)
B, Tt, Tp, Ts = 8, 32, 150, 250
dev = 'cuda'
text = torch.randint(0, cfg.text_vocab, (B, Tt), device=dev, generator=g); tlen = torch.full((B,), Tt, dtype=torch.int32, device=dev)
spk = torch.randn(B, cfg.spk_dim, device=dev, generator=g)
style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g)
u = torch.rand(Ts, B, 2, device=dev, generator=g)
lm = eng.lm
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(4):
    torch.cuda.synchronize(); h0 = time.perf_counter()
    e0 = ev(); pre = lm.prefix(text, tlen, spk, style_tok); h1 = time.perf_counter(); e1 = ev()
    pr, ks = pre, None
    cache = lm.new_cache(B, pr.shape[0] + Ts); e2 = ev()
    hid = lm.forward_new(pr, cache, 0, ks); h2 = time.perf_counter(); e3 = ev()
    lg = lm.logits(hid[-1]).contiguous(); e4 = ev()
    toks = lm.decode(pr, Ts, u, True); e5 = ev()
    toks1 = lm.decode(pr, 2, u[:2].contiguous(), True); e6 = ev()
    torch.cuda.synchronize()
    print(f'iter {it}: prefix {e0.elapsed_time(e1):.2f} ms (host {1e3 * (h1 - h0):.2f}), cache alloc {e1.elapsed_time(e2):.2f}, prefill {e2.elapsed_time(e3):.2f} (host {1e3 * (h2 - h1):.2f}), logits0 {e3.elapsed_time(e4):.2f}; '
          f'whole decode call {e4.elapsed_time(e5):.2f}; 2-step decode call {e5.elapsed_time(e6):.2f}; prefix rows {pr.shape[0]}')
