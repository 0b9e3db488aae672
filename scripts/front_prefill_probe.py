"""Pipelined step with the LM prefill on the decode chain (default) vs on the caller's front stream, alternating in one process."""
import sys, time
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
def run(pipe, steps=12):
    outs = []
    with torch.cuda.stream(pipe.front_stream):
        for _ in range(4): pipe.submit(*args)
        pipe.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = pipe.submit(*args)
            if r is not None: outs.append(r)
        outs += pipe.drain(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, outs
pipes = {fp: PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, pipe_classes=classes, front_prefill=fp) for fp in (False, True)}
ref = None
for rep in range(4):
    for fp in (False, True):
        ms, outs = run(pipes[fp])
        toks = outs[-1][0]
        if ref is None: ref = toks.clone()
        print(f'rep {rep} front_prefill={fp}: {ms:.2f} ms/batch  tokens equal: {bool((toks == ref).all())}', flush=True)
for cob in (2,):
    pc = {fp: PipelinedSynth(eng, lm_depth=2, lm_priority=0, render_priority=0, pipe_classes=classes, front_prefill=fp, cobatch=cob) for fp in (False, True)}
    for rep in range(3):
        for fp in (False, True):
            ms, outs = run(pc[fp])
            print(f'cobatch {cob} rep {rep} front_prefill={fp}: {ms:.2f} ms/batch  tokens equal: {bool((outs[-1][0] == ref).all())}', flush=True)
