"""Decode engines side by side at BASELINE config 2's shapes: 250 steps x b rows alone on the device (ms per decode call), per engine.
LM_TIME_ROWS=8,16,32  LM_TIME_ENGINES=v2,v3  LM_TIME_DBG=0,1,2,... (ASTTS_LM_FUSED_DBG values tried with v3: timing only, wrong results)"""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.synth.config import SynthConfig
from astts.synth.model import AcousticLM
from astts.synth.weights import make_lm_weights
dev = torch.device('cuda', 0)
cfg = SynthConfig()
lm = AcousticLM(make_lm_weights(cfg, 0), cfg, dev)
rows = [int(x) for x in os.environ.get('LM_TIME_ROWS', '8,16,32').split(',')]
engines = os.environ.get('LM_TIME_ENGINES', 'v2,v3,v2,v3').split(',')
dbgs = os.environ.get('LM_TIME_DBG', '0').split(',')
for b in rows:
    g = torch.Generator().manual_seed(b)
    tt, tp, steps = 32, 150, 250
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g).to(dev)
    tlen = torch.full((b,), tt, dtype=torch.int32, device=dev)
    spk = torch.randn(b, cfg.spk_dim, generator=g).to(dev)
    prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g).to(dev)
    u = torch.rand(steps, b, 2, generator=g).to(dev)
    forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g).to(dev)
    pre = lm.prefix(text, tlen, spk, prompt)
    for eng in engines:
        for dbg in (dbgs if eng == 'v3' else ['0']):
            os.environ['ASTTS_LM_ENGINE'] = eng
            os.environ['ASTTS_LM_FUSED_DBG'] = dbg
            lm.decode(pre, steps, u, True, forced); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): lm.decode(pre, steps, u, True, forced)
            torch.cuda.synchronize()
            print(f'b={b} {eng} dbg={dbg}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per 250-step decode', flush=True)
