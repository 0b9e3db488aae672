#!/bin/bash
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ras_sample" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python -m pytest tests/test_lm_step_gpu.py tests/test_synth_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python3 - <<'PY'
import sys, time, torch
sys.path[:0] = ['.', 'autostyle-tts_amd']
from astts.synth.config import SynthConfig
from astts.synth.model import AcousticLM
from astts.synth.weights import make_lm_weights
for pol in ('mask', 'reject'):
    cfg = SynthConfig(eos_policy=pol); lm = AcousticLM(make_lm_weights(cfg, 0), cfg, torch.device('cuda'))
    g = torch.Generator().manual_seed(3); b, tt, tp = 8, 32, 150
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g).cuda(); tlen = torch.full((b,), tt, dtype=torch.int32).cuda()
    spk = torch.randn(b, cfg.spk_dim, generator=g).cuda(); prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g).cuda()
    pre = lm.prefix(text, tlen, spk, prompt); u = torch.rand(250, b, 2, device='cuda')
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); lm.decode(pre, 250, u, True, None); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f'eos_policy {pol}: 250 decode steps {best * 1e3:.1f} ms')
PY
