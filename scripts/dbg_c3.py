import sys, math, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.model import SynthEngine
from astts.synth.weights import make_all
DEV = 'cuda'
cfg = SynthConfig()
eng = SynthEngine(make_all(cfg, 0), cfg, DEV)
g = torch.Generator(device=DEV).manual_seed(3)
B, Tt, Tp, Ts = 64, 64, 150, int(os.environ.get('TS', '60'))
text = torch.randint(0, cfg.text_vocab, (B, Tt), device=DEV, generator=g)
tlen = torch.full((B,), Tt, dtype=torch.int32, device=DEV)
spk_s = torch.randn(B, cfg.spk_dim, device=DEV, generator=g)
style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=DEV, generator=g)
u = torch.rand(Ts, B, 2, device=DEV, generator=g)
sl = slice(32, 64)
pre64 = eng.lm.prefix(text, tlen, spk_s, style_tok)
pre32 = eng.lm.prefix(text[sl], tlen[sl], spk_s[sl], style_tok[sl])
print('prefix equal:', torch.equal(pre64[:, sl], pre32), float((pre64[:, sl] - pre32).abs().max()))
t64 = eng.lm.decode(pre64, Ts, u, True)
t32 = eng.lm.decode(pre64[:, sl].contiguous(), Ts, u[:, sl].contiguous(), True)
t32b = eng.lm.decode(pre64[:, sl].contiguous(), Ts, u[:, sl].contiguous(), True)
t32c = eng.lm.decode(pre32, Ts, u[:, sl].contiguous(), True)
torch.cuda.synchronize()
print('decode 64-batch rows 32.. vs own batch (same prefix values):', torch.equal(t64[sl], t32), 'repeat:', torch.equal(t32, t32b), 'own prefix:', torch.equal(t32c, t32))
if not torch.equal(t64[sl], t32):
    d = (t64[sl] != t32)
    print('first differing step per row:', [int(r.nonzero()[0]) if r.any() else -1 for r in d][:32])
t64b = eng.lm.decode(pre64, Ts, u, True)
print('64-batch repeat equal:', torch.equal(t64, t64b))
