"""Where the time of a ragged test-set shard goes (BASELINE config 4, one rank's 203 sentences through the drop-in batch surface):
the engine stages are wrapped with synchronising timers (so the numbers add up; the run itself is slower than the unwrapped one)."""
import json, math, os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch, warnings
from astts.compat.cosyvoice import CosyVoice
warnings.simplefilter("ignore")
sents = json.load(open('tests/golden/iemocap_test_sentences.json'))['all'][:int(os.environ.get('N', 203))]
want = [int(min(max(round(20 * len(x.split())), 25), 1500)) for x in sents]
cv = CosyVoice('/nonexistent', seed=0, allow_random_init=True)
g = torch.Generator().manual_seed(0)
t16 = torch.arange(int(2.5 * 16000)) / 16000
style = (0.3 * torch.sin(2 * math.pi * 220 * t16) + 0.01 * torch.randn(t16.shape, generator=g))[None]
timbre = (0.3 * torch.sin(2 * math.pi * 330 * t16[:32000]) + 0.01 * torch.randn(32000, generator=g))[None]
items = [(x, 'He did. In Niagara Falls.', style, timbre) for x in sents]
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        T[key] = T.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = cv.inference_tts_with_st_batch(items, max_batch=int(os.environ.get('MB', 32)), split=False, fixed_tokens=want, seeds=list(range(len(items))))
    torch.cuda.synchronize(); return time.perf_counter() - t0, sum(o[0]['tts_speech'].shape[1] for o in outs) / cv.cfg.sample_rate
run()
dt, audio = run()
print(f'unwrapped: {audio:.0f} s of audio in {dt:.2f} s = {audio / dt:.0f}x')
eng = cv.engine
wrap(eng.lm, 'prefix_ragged', 'lm prefix'); wrap(eng.lm, 'decode', 'lm decode'); wrap(eng.flow, 'decode_ragged', 'flow'); wrap(eng.hift, 'forward', 'vocoder')
wrap(cv, '_draws', 'draws'); wrap(cv.frontend, 'prompt', 'prompts')
dt, audio = run()
print(f'wrapped: {dt:.2f} s;', {k: round(v, 2) for k, v in T.items()}, 'rest', round(dt - sum(T.values()), 2))
