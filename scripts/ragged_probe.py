"""Config-4-style slice on one GPU: the first N IEMOCAP test sentences (tests/golden), ragged lengths, through the drop-in
CosyVoice.inference_tts_with_st_batch surface (byte tokenizer + stand-in frontend, random-init full-size weights)."""
import sys, time, json, os, math, warnings
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
warnings.simplefilter('ignore')
from astts.compat.cosyvoice import CosyVoice
N = int(os.environ.get('N', '64')); BS = int(os.environ.get('BS', '32'))
sents = json.load(open('tests/golden/iemocap_test_sentences.json'))
if isinstance(sents, dict): sents = sents.get('sentences') or list(sents.values())[0]
texts = [s if isinstance(s, str) else s.get('text', str(s)) for s in sents][:N]
cv = CosyVoice('/nonexistent', seed=0, allow_random_init=True)
sr = 16000
t = torch.arange(int(2.5 * sr)) / sr
style = (0.3 * torch.sin(2 * math.pi * 220 * t) + 0.01 * torch.randn(t.shape))[None]
timbre = (0.3 * torch.sin(2 * math.pi * 330 * t[: 2 * sr]) + 0.01 * torch.randn(2 * sr))[None]
items = [(tx, "He did. In Niagara Falls.", style, timbre) for tx in texts]
cv.inference_tts_with_st_batch(items[:4], max_batch=4)      # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = cv.inference_tts_with_st_batch(items, max_batch=BS)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
audio = sum(seg['tts_speech'].shape[1] for o in outs for seg in o) / cv.sample_rate
nseg = sum(len(o) for o in outs)
ok = all(bool(torch.isfinite(seg['tts_speech']).all()) for o in outs for seg in o)
lens = sorted(seg['tts_speech'].shape[1] / cv.sample_rate for o in outs for seg in o)
print(f'{len(items)} utterances ({nseg} segments, batch {BS}): {audio:.1f} s of audio in {dt:.2f} s = RTF^-1 {audio / dt:.1f}; finite {ok}; '
      f'segment seconds min/median/max {lens[0]:.1f}/{lens[len(lens)//2]:.1f}/{lens[-1]:.1f}')
if os.environ.get('PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    cv.inference_tts_with_st_batch(items, max_batch=BS); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
