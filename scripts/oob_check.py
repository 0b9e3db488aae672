"""Guard-band every output the operator wrappers allocate (torch.empty / empty_like / zeros patched) and check
that no kernel writes outside its tensor.  Runs the LM prefix, the flow operator path and the vocoder at full size."""
import sys, math, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_all
from astts.synth.model import SynthEngine
from astts import ops
cfg = SynthConfig() if os.environ.get('TINY') != '1' else SynthConfig.tiny()
W = make_all(cfg, 0); eng = SynthEngine(W, cfg, 'cuda'); del W
G = 4096
PAT = 0x5a
records = []
_empty, _empty_like, _zeros = torch.empty, torch.empty_like, torch.zeros
import traceback
def guarded(shape, dtype, device, fill=None):
    if isinstance(shape, int): shape = (shape,)
    n = 1
    for s_ in shape: n *= int(s_)
    isz = torch.tensor([], dtype=dtype).element_size()
    nbytes = n * isz
    pad = (-nbytes) % 16
    buf = torch.full((G + nbytes + pad + G,), PAT, dtype=torch.uint8, device=device)
    view = buf[G:G + nbytes].view(dtype).view(*shape)
    if fill is not None: view.fill_(fill)
    where = ''.join(traceback.format_stack(limit=4)[:-2]).strip().split('\n')[-2:]
    records.append((buf, nbytes, tuple(shape), dtype, where))
    return view
def p_empty(*size, dtype=None, device=None, **kw):
    if device is None or 'cuda' not in str(device): return _empty(*size, dtype=dtype, device=device, **kw)
    shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
    return guarded(tuple(shape), dtype or torch.float32, device)
def p_empty_like(t, **kw):
    if not t.is_cuda: return _empty_like(t, **kw)
    return guarded(tuple(t.shape), kw.get('dtype', t.dtype), t.device)
def check(tag):
    bad = 0
    for buf, nbytes, shape, dtype, where in records:
        lo = buf[:G]; hi = buf[G + nbytes:]
        if not bool((lo == PAT).all()) or not bool((hi == PAT).all()):
            bad += 1
            nlo = int((lo != PAT).sum()); nhi = int((hi != PAT).sum())
            first_hi = int((hi != PAT).nonzero()[0]) if nhi else -1
            print(f'  OOB WRITE: tensor {shape} {dtype}: {nlo} bytes before, {nhi} bytes after (first at +{first_hi}); allocated at', where)
    print(f'{tag}: {len(records)} guarded tensors, {bad} with out-of-bounds writes', flush=True)
    records.clear()
torch.empty, torch.empty_like = p_empty, p_empty_like
g = torch.Generator(device='cuda').manual_seed(0)
dev = 'cuda'
B, Tt, Tp, Ts = 8, 32, 150, int(os.environ.get('TS', '40'))
text = torch.randint(0, cfg.text_vocab, (B, Tt), device=dev, generator=g); tlen = torch.full((B,), Tt, dtype=torch.int32, device=dev)
spk_s = torch.randn(B, cfg.spk_dim, device=dev, generator=g); spk_t = torch.randn(B, cfg.spk_dim, device=dev, generator=g)
style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g); timbre_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g)
tmp = cfg.mel_frames_for_tokens(Tp); tm = cfg.mel_frames_for_tokens(Ts)
timbre_mel = torch.randn(B, tmp, cfg.mel, device=dev, generator=g)
u = torch.rand(Ts, B, 2, device=dev, generator=g); z = torch.randn(B, tmp + tm, cfg.mel, device=dev, generator=g)
nh = cfg.nb_harmonics + 1
phase0 = (torch.rand(B, nh, device=dev, generator=g) * 2 - 1) * math.pi; phase0[:, 0] = 0
noise = torch.randn(B, tm * cfg.upsample_total, nh, device=dev, generator=g)
pre = eng.lm.prefix(text, tlen, spk_s, style_tok); torch.cuda.synchronize(); check('lm.prefix')
toks = eng.lm.decode(pre, Ts, u, True); torch.cuda.synchronize(); check('lm.decode (engine)')
toks2 = eng.lm.decode(pre, 6, u[:6], True, use_engine=False) if 'use_engine' in eng.lm.decode.__code__.co_varnames else None
torch.cuda.synchronize(); check('lm.decode (python steps)')
all_tok = torch.cat([timbre_tok.to(torch.int32), toks], 1); tl = torch.full((B,), all_tok.shape[1], dtype=torch.int32, device=dev)
eng.flow.use_engine = False
mel = eng.flow.decode(all_tok, tl, timbre_mel, spk_t, z, tmp + tm); torch.cuda.synchronize(); check('flow.decode (operator path)')
eng.flow.use_engine = True
mel2 = eng.flow.decode(all_tok, tl, timbre_mel, spk_t, z, tmp + tm); torch.cuda.synchronize(); check('flow.decode (engine)')
print('engine == ops:', bool(torch.equal(mel, mel2)))
wav = eng.hift.forward(mel, phase0, noise); torch.cuda.synchronize(); check('hift.forward')
# ragged flow
lens = [(12, 20), (5, 33), (21, 8)]
toks_l = [torch.randint(0, cfg.speech_vocab, (a + b_,), generator=torch.Generator().manual_seed(a)) for a, b_ in lens]
pm = [torch.randn(cfg.mel_frames_for_tokens(a), cfg.mel) for a, _ in lens]
zs = [torch.randn(cfg.mel_frames_for_tokens(a) + cfg.mel_frames_for_tokens(b_), cfg.mel) for a, b_ in lens]
eng.flow.use_engine = False
eng.flow.decode_ragged(toks_l, pm, torch.randn(3, cfg.spk_dim), zs); torch.cuda.synchronize(); check('flow.decode_ragged (operator path)')
