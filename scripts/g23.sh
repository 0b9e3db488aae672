#!/bin/bash
# the wide (32-column) decode GEMV form: parity with it switched on, chain benchmark numbers, then A/B inside the pipelined benchmark
ASTTS_LM_WIDE=1 python -m pytest tests/test_lm_step_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python3 - <<'PY'
import os, sys, subprocess
sys.path[:0] = ['.', 'autostyle-tts_amd']
code = '''
import sys, torch
sys.path[:0] = ['.', 'autostyle-tts_amd']
from astts.synth.config import SynthConfig
from astts.synth.model import AcousticLM
from astts.synth.weights import make_lm_weights
cfg = SynthConfig(); lm = AcousticLM(make_lm_weights(cfg, 0), cfg, torch.device('cuda'))
g = torch.Generator().manual_seed(3); b, tt, tp, steps = 8, 32, 150, 40
text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g).cuda(); tlen = torch.full((b,), tt, dtype=torch.int32).cuda()
spk = torch.randn(b, cfg.spk_dim, generator=g).cuda(); prompt = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g).cuda()
u = torch.rand(steps, b, 2, generator=g).cuda(); forced = torch.randint(0, cfg.speech_vocab, (b, steps), generator=g).cuda()
pre = lm.prefix(text, tlen, spk, prompt)
toks, lg = lm.decode(pre, steps, u, True, forced, return_logits=True)
free = lm.decode(pre, steps, u, True, None)
torch.save((lg.cpu(), free.cpu()), sys.argv[1])
import time
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); lm.decode(pre, 250, torch.rand(250, b, 2, device='cuda'), True, None); torch.cuda.synchronize()
print('decode 250 steps: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
'''
for w in ('0', '1'):
    env = dict(os.environ, ASTTS_LM_WIDE=w)
    r = subprocess.run([sys.executable, '-c', code, f'/tmp/wide{w}.pt'], env=env, capture_output=True, text=True)
    print('ASTTS_LM_WIDE=' + w, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
import torch
a, b = torch.load('/tmp/wide0.pt'), torch.load('/tmp/wide1.pt')
print('logits bit-identical:', torch.equal(a[0], b[0]), 'tokens identical:', torch.equal(a[1], b[1]))
PY
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['stages_ms']['lm_ms'], d['pipelining'][:14])"); echo "$*: $v"; }
for rep in 1 2; do run ASTTS_LM_WIDE=0; run ASTTS_LM_WIDE=1; done
