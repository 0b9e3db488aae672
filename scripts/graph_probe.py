"""Does capturing the whole fixed-shape synthesis step into one hipGraph (torch.cuda.CUDAGraph) pay?"""
import sys, time, math
sys.path[:0] = ['.', 'autostyle-tts_amd', 'scripts']
import torch
from astts.synth.config import SynthConfig
from astts.synth.weights import make_all
from astts.synth.model import SynthEngine
cfg = SynthConfig(); W = make_all(cfg, 0); eng = SynthEngine(W, cfg, 'cuda'); del W
g = torch.Generator(device='cuda').manual_seed(0)
B, Tt, Tp, Ts = 8, 32, 150, 250
dev = 'cuda'
text = torch.randint(0, cfg.text_vocab, (B, Tt), device=dev, generator=g); tlen = torch.full((B,), Tt, dtype=torch.int32, device=dev)
spk_s = torch.randn(B, cfg.spk_dim, device=dev, generator=g); spk_t = torch.randn(B, cfg.spk_dim, device=dev, generator=g)
style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g); timbre_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=dev, generator=g)
tmp = cfg.mel_frames_for_tokens(Tp); tm = cfg.mel_frames_for_tokens(Ts)
timbre_mel = torch.randn(B, tmp, cfg.mel, device=dev, generator=g)
u = torch.rand(Ts, B, 2, device=dev, generator=g); z = torch.randn(B, tmp + tm, cfg.mel, device=dev, generator=g)
nh = cfg.nb_harmonics + 1
phase0 = (torch.rand(B, nh, device=dev, generator=g) * 2 - 1) * math.pi; phase0[:, 0] = 0
noise = torch.randn(B, tm * cfg.upsample_total, nh, device=dev, generator=g)
def run():
    return eng.tts(text, tlen, spk_s, style_tok, Ts, u, timbre_tok, timbre_mel, spk_t, z, phase0, noise)
for _ in range(2): toks0, mel0, wav0 = run()
torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print('eager step %.1f ms' % ((time.perf_counter() - t0) * 1e3))
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream().wait_stream(s)
t0 = time.perf_counter()
with torch.cuda.graph(graph):
    toks, mel, wav = run()
torch.cuda.synchronize(); print('capture %.1f s' % (time.perf_counter() - t0))
for _ in range(2): graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): graph.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print('graph replay step %.1f ms  RTF^-1 %.1f' % (dt * 1e3, B * wav.shape[1] / cfg.sample_rate / dt))
print('same tokens', bool(torch.equal(toks, toks0)), 'wav max diff', float((wav - wav0).abs().max()))
