"""Exclusive-phase schedule experiment: G decode chains (G batches) run concurrently with nothing else on the GPU, then the G
batches are rendered one after the other with nothing else on the GPU.  Compare with the overlapped pipeline (bench.py)."""
import sys, time, threading, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
import bench
from astts import ops
from astts.synth.config import SynthConfig
from astts.synth.model import SynthEngine
from astts.synth.weights import make_all
cfg = SynthConfig()
dev = torch.device('cuda')
eng = SynthEngine(make_all(cfg, 0), cfg, dev)
inp = bench.SynthInputs(cfg, 8, 32, 150, 250, dev, seed=100)
classes = ops.stream_pipe_classes(device=dev, verbose=True)
firsts = [c[0] for c in classes]


def lm(stream, out, i):
    with torch.cuda.device(dev), torch.cuda.stream(stream):
        pre = eng.lm.prefix(inp.text, inp.tlen, inp.spk_style, inp.style_tok)
        out[i] = eng.lm.decode(pre, inp.ts, inp.u, ignore_eos=True)
        stream.synchronize()


def render(toks):
    return eng.tts_render(toks, inp.timbre_tok, inp.timbre_mel, inp.spk_timbre, inp.z, inp.phase0, inp.noise)


for G in (1, 2, 3, 4):
    streams = firsts[:G]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = [None] * G
        th = [threading.Thread(target=lm, args=(streams[i], out, i)) for i in range(G)]
        for t in th: t.start()
        for t in th: t.join()
        t1 = time.perf_counter()
        with torch.cuda.stream(firsts[0]):
            for i in range(G):
                render(out[i])
            firsts[0].synchronize()
        t2 = time.perf_counter()
    print(f'G={G}: LM phase {(t1 - t0) * 1e3:.1f} ms ({(t1 - t0) * 1e3 / G:.1f} per batch), render phase {(t2 - t1) * 1e3:.1f} ms '
          f'({(t2 - t1) * 1e3 / G:.1f} per batch) -> {(t2 - t0) * 1e3 / G:.1f} ms per batch = {inp.audio_seconds / ((t2 - t0) / G):.0f}x')
