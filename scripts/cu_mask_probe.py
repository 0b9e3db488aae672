"""Pipelined step with the RENDER stream restricted to a subset of the CUs (decode chains unrestricted): does leaving CUs free for
the decode chains beat sharing the whole chip?  Config-2 shapes, same inputs as bench.py."""
import os, sys, time, math
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
base = PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, pipe_classes=classes)
print(f'unmasked (pipe-class streams): {run(base):.1f} ms/batch, again {run(base):.1f}', flush=True)
def masks(kind, n):
    if kind == 'low': return list(range(n))
    if kind == 'spread': return [i for i in range(256) if (i * n) // 256 != ((i - 1) * n) // 256 or i == 0][:n]
    if kind == 'xcd': return [i for i in range(256) if (i % 8) < n * 8 // 256]      # whole XCDs if CU ids interleave over XCDs
for kind in ('low', 'spread', 'xcd'):
    for n in (224, 192, 160, 128):
        try:
            s_r = ops.cu_masked_stream(masks(kind, n), dev)
            lm = [c[0] for c in classes[1:3]] if len(classes) >= 3 else [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            pipe = PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, streams=lm + [s_r])
            # render alone on the masked stream
            with torch.cuda.stream(s_r):
                toks = eng.tts_tokens(*args[:6]); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s_r); eng.tts_render(toks, *args[6:]); e1.record(s_r); torch.cuda.synchronize()
            print(f'render mask {kind} {n} CUs: render alone {e0.elapsed_time(e1):.1f} ms; pipelined {run(pipe):.1f} ms/batch, again {run(pipe):.1f}', flush=True)
        except Exception as e:
            print('mask', kind, n, 'failed:', repr(e)[:200], flush=True)
