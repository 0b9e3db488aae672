"""Complementary CU masks: the render stream on the LOW n mask bits, both decode chains on the HIGH 256 - n bits.
(scripts/micro/cu_census.hip: mask bit b enables one CU of XCC b % 8, so 'low n' = n / 8 CUs of every XCC -- whole XCCs cannot be
taken away from a queue.)  Prints the render stage alone, one decode alone, and the pipelined step per partition, beside the
unmasked pipeline of the same process.  Config-2 shapes, same inputs as bench.py."""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
import bench
from astts import ops
from astts.synth.config import SynthConfig
from astts.synth.model import PipelinedSynth, SynthEngine
from astts.synth.weights import make_all
dev = torch.device('cuda', 0)
cfg = SynthConfig()
eng = SynthEngine(make_all(cfg, 0), cfg, dev)
inp = bench.SynthInputs(cfg, 8, 32, 150, 250, dev, seed=100)
args = (inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok, inp.timbre_mel, inp.spk_timbre, inp.z, inp.phase0, inp.noise)
classes = ops.stream_pipe_classes(device=dev)
print('pipe classes:', [len(c) for c in classes], flush=True)


def run(pipe, steps=10):
    with torch.cuda.stream(pipe.front_stream):
        for _ in range(4): pipe.submit(*args)
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): pipe.submit(*args)
        pipe.drain(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def alone(s_r, s_c):
    with torch.cuda.stream(s_c):
        toks = eng.tts_tokens(*args[:6]); torch.cuda.synchronize()
        t0 = time.perf_counter(); toks = eng.tts_tokens(*args[:6]); torch.cuda.synchronize()
        lm = (time.perf_counter() - t0) * 1e3
    with torch.cuda.stream(s_r):
        eng.tts_render(toks, *args[6:]); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.tts_render(toks, *args[6:]); torch.cuda.synchronize()
        rd = (time.perf_counter() - t0) * 1e3
    return lm, rd


base = PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, pipe_classes=classes)
lm0, rd0 = alone(base.s_render, base.s_lm[0])
print(f'unmasked: decode alone {lm0:.1f} ms, render alone {rd0:.1f} ms; pipelined {run(base):.1f} ms/batch, again {run(base):.1f}', flush=True)
for n_chain, share in (() if os.environ.get('PROBE2_ONLY_CHAINS') else ((32, True), (48, True), (64, True), (64, False), (96, True), (128, True), (0, True))):
    try:
        n_r = 256 - n_chain if n_chain else 192
        s_r = ops.cu_masked_stream(list(range(n_r)), dev)
        if n_chain == 0:            # control: render masked to 192, chains unmasked
            lm = [c[0] for c in classes[1:3]]
        elif share:
            lm = [ops.cu_masked_stream(list(range(n_r, 256)), dev) for _ in range(2)]
        else:                       # each chain its own half of the chains' CUs
            h = n_chain // 2
            lm = [ops.cu_masked_stream(list(range(n_r, n_r + h)), dev), ops.cu_masked_stream(list(range(n_r + h, 256)), dev)]
        pipe = PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, streams=lm + [s_r])
        l1, r1 = alone(s_r, lm[0])
        print(f'render low {n_r} / chains high {n_chain} ({"shared" if share else "split"}): decode alone {l1:.1f} ms, render alone {r1:.1f} ms; '
              f'pipelined {run(pipe):.1f} ms/batch, again {run(pipe):.1f}', flush=True)
    except Exception as e:
        print('partition', n_chain, 'failed:', repr(e)[:300], flush=True)

# chains confined to the HIGH n mask bits, render unrestricted: decode workgroups then never sit on the other CUs, so the render
# stage's one-workgroup-per-CU kernels are only delayed on the chains' share of the chip
for n_chain in (224, 192, 160, 128, 96):
    try:
        lm = [ops.cu_masked_stream(list(range(256 - n_chain, 256)), dev) for _ in range(2)]
        pipe = PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, streams=lm + [classes[0][0]])
        l1, r1 = alone(classes[0][0], lm[0])
        print(f'render unmasked / chains high {n_chain} (shared): decode alone {l1:.1f} ms, render alone {r1:.1f} ms; '
              f'pipelined {run(pipe):.1f} ms/batch, again {run(pipe):.1f}', flush=True)
    except Exception as e:
        print('chains-only mask', n_chain, 'failed:', repr(e)[:300], flush=True)
