"""What the pipelined step is sensitive to: the same pipeline (three decode chains, 4 hardware queues) with 10 % less render work (9 Euler
steps instead of 10) or 10 % fewer decode steps (225 instead of 250).  Sensitivity probe only (the outputs are not the benchmark's)."""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
import bench
from astts import ops
from astts.synth.config import SynthConfig
from astts.synth.model import PipelinedSynth, SynthEngine
from astts.synth.weights import make_all
dev = torch.device('cuda', 0)
W = make_all(SynthConfig(), 0)
classes = None
def run(cfm_steps, ts, depth=3, steps=16):
    global classes
    cfg = SynthConfig(cfm_steps=cfm_steps)
    eng = SynthEngine(W, cfg, dev)
    inp = bench.SynthInputs(cfg, 8, 32, 150, ts, dev, seed=100)
    args = (inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok, inp.timbre_mel, inp.spk_timbre, inp.z, inp.phase0, inp.noise)
    if classes is None:
        classes = ops.stream_pipe_classes(device=dev)
    pipe = PipelinedSynth(eng, lm_depth=depth, lm_priority=0, render_priority=0, pipe_classes=classes)
    out = []
    for _ in range(2):
        with torch.cuda.stream(pipe.front_stream):
            for _ in range(4): pipe.submit(*args)
            pipe.drain(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps): pipe.submit(*args)
            pipe.drain(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    # stage times alone
    torch.cuda.synchronize(); t0 = time.perf_counter(); toks = eng.tts_tokens(*args[:6]); torch.cuda.synchronize(); lm = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); eng.tts_render(toks, *args[6:]); torch.cuda.synchronize(); rd = (time.perf_counter() - t0) * 1e3
    del pipe, eng
    return out, lm, rd
for name, c, t in (('baseline', 10, 250), ('render -10 %', 9, 250), ('decode -10 %', 10, 225), ('baseline', 10, 250), ('render -20 %', 8, 250), ('decode -20 %', 10, 200)):
    o, lm, rd = run(c, t)
    print(f'{name:14s} Euler steps {c:2d}, decode steps {t}: pipelined {o[0]:.2f} / {o[1]:.2f} ms per batch; alone: LM {lm:.1f} ms, render {rd:.1f} ms', flush=True)
