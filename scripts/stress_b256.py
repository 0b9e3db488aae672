"""Determinism stress of the wide-batch decode: B rows decoded as 32-row groups on two concurrent streams, several times; every
run must give the same tokens, and rows 0..7 / the last 8 rows must equal their own 8-row run."""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.model import AcousticLM
from astts.synth.weights import make_lm_weights
cfg = SynthConfig()
lm = AcousticLM(make_lm_weights(cfg, 0), cfg, torch.device('cuda'))
B, Tt, Tp, Ts = int(os.environ.get('B', '256')), 32, 150, int(os.environ.get('TS', '250'))
g = torch.Generator(device='cuda').manual_seed(5)
text = torch.randint(0, cfg.text_vocab, (B, Tt), device='cuda', generator=g)
tlen = torch.full((B,), Tt, dtype=torch.int32, device='cuda')
spk = torch.randn(B, cfg.spk_dim, device='cuda', generator=g)
ptok = torch.randint(0, cfg.speech_vocab, (B, Tp), device='cuda', generator=g)
u = torch.rand(Ts, B, 2, device='cuda', generator=g)
pre = lm.prefix(text, tlen, spk, ptok)
ref = None
bad = 0
for it in range(int(os.environ.get('ITERS', '6'))):
    t0 = time.time()
    toks = lm.decode(pre, Ts, u, ignore_eos=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = toks.clone()
    diff = (toks != ref)
    t8 = lm.decode(pre[:, :8].contiguous(), Ts, u[:, :8].contiguous(), ignore_eos=True)
    tl = lm.decode(pre[:, B - 8:].contiguous(), Ts, u[:, B - 8:].contiguous(), ignore_eos=True)
    d8, dl = (t8 != toks[:8]), (tl != toks[B - 8:])
    rows = sorted(set(diff.nonzero()[:, 0].tolist()))
    print(f'iter {it}: {time.time() - t0:.2f} s; rows differing from run 0: {rows[:20]} ({len(rows)}); first step {int(diff.nonzero()[:, 1].min()) if len(rows) else -1}; '
          f'rows 0..7 vs batch-8: {int(d8.any(1).sum())} differ; last 8: {int(dl.any(1).sum())} differ', flush=True)
    bad += len(rows) + int(d8.any()) + int(dl.any())
print('BAD' if bad else 'OK')
