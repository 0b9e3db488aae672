"""The stream=True state machine (astts.synth.stream: upstream CosyVoiceModel.tts(stream=True) / token2wav restated) on toy
stage functions: chunk boundaries, withheld overlaps, cross-fade arithmetic, single-chunk == one-shot."""
import torch

from astts.synth.config import SynthConfig
from astts.synth.stream import StreamConsts, hamming, stream_render


def _stages(cfg, log):
    frames = cfg.mel_frames_for_tokens

    def flow_mel(tok):
        log.append(int(tok.numel()))
        n = frames(int(tok.numel()))
        # frame j of a chunk starting at token value t0 = t0 + j / n: distinguishable, smooth
        t0 = float(tok[0])
        return (t0 + torch.arange(n, dtype=torch.float32) / max(n, 1)).view(1, n, 1).repeat(1, 1, cfg.mel)

    def f0(mel):
        return mel[..., 0]

    def source(f):
        return f.repeat_interleave(cfg.hop, dim=1) * 1e-3

    def vocode(mel, src):
        return mel[..., 0].repeat_interleave(cfg.hop, dim=1) * 1e-2 + src

    return flow_mel, f0, source, vocode


def test_constants_follow_upstream():
    c = StreamConsts.for_config(SynthConfig())
    assert (c.token_min_hop, c.token_max_hop, c.token_overlap, c.mel_overlap, c.mel_cache, c.source_cache) == (100, 200, 20, 34, 20, 5120)
    w = hamming(68)
    assert abs(float(w[0]) - 0.08) < 1e-6 and abs(float(w[-1]) - 0.08) < 1e-6 and float(w.max()) <= 1.0
    assert torch.allclose(w, torch.tensor(__import__("numpy").hamming(68), dtype=torch.float32), atol=1e-7)


def test_chunking_lengths_and_single_chunk_equals_one_shot():
    cfg = SynthConfig()
    c = StreamConsts.for_config(cfg)
    # fewer than hop + overlap tokens: ONE chunk, identical to the one-shot render
    log = []
    fm, f0, src, voc = _stages(cfg, log)
    toks = torch.arange(80)
    chunks = list(stream_render(toks, c, fm, f0, src, voc))
    mel = fm(toks)
    assert len(chunks) == 1 and torch.equal(chunks[0], voc(mel, src(f0(mel))))
    # 330 tokens: hops at 0, 100, 200 (120 tokens each), then the 30 left over as the final chunk
    log.clear()
    chunks = list(stream_render(torch.arange(330), c, fm, f0, src, voc))
    assert log == [120, 120, 120, 30] and len(chunks) == 4
    n120, n30 = cfg.mel_frames_for_tokens(120), cfg.mel_frames_for_tokens(30)
    first = (n120 - c.mel_overlap) * cfg.hop - c.source_cache
    mid = (c.mel_cache + n120 - c.mel_overlap) * cfg.hop - c.source_cache
    last = (c.mel_cache + n30) * cfg.hop
    assert [int(x.shape[1]) for x in chunks] == [first, mid, mid, last]
    assert all(bool(torch.isfinite(x).all()) for x in chunks)
    # emitted audio per 100-token hop ~ 2 s (172 frames of the 172.27 a hop spans)
    assert abs(mid / cfg.sample_rate - 2.0) < 0.01


def test_mel_crossfade_arithmetic():
    """Second chunk's first mel_overlap frames = new * w[:L] + withheld tail of the first chunk * w[L:]."""
    cfg = SynthConfig()
    c = StreamConsts.for_config(cfg)
    seen = []
    log = []
    fm, f0, src, _ = _stages(cfg, log)

    def voc(mel, s):
        seen.append(mel.clone())
        return torch.zeros(1, mel.shape[1] * cfg.hop)

    list(stream_render(torch.arange(240), c, fm, f0, src, voc))
    L, w = c.mel_overlap, hamming(2 * c.mel_overlap)
    m0, m1 = fm(torch.arange(120)), fm(torch.arange(100, 220))
    assert torch.equal(seen[0], m0[:, :-L])
    head = m1[:, :L] * w[:L].view(1, L, 1) + m0[:, -L:] * w[L:].view(1, L, 1)
    # the vocoder's second input = 20 cached frames of the first (cut) mel + the faded second mel, its own tail withheld
    assert torch.equal(seen[1][:, :c.mel_cache], m0[:, :-L][:, -c.mel_cache:])
    assert torch.allclose(seen[1][:, c.mel_cache:c.mel_cache + L], head, atol=1e-6)
    assert torch.equal(seen[1][:, c.mel_cache + L:], m1[:, L:-L])


class _GrowingTokens:
    """A token source as `stream_render` takes it in place of a finished tensor (the LM still decoding: compat.cosyvoice._LmTokenStream):
    tokens become available `grow` at a time; `wait(n)` returns what exists once at least n do (or everything, finished)."""

    def __init__(self, tokens, grow):
        self.tokens, self.grow, self.have, self.calls = tokens, grow, 0, []

    def wait(self, n):
        while self.have < min(n, self.tokens.numel()):
            self.have = min(self.have + self.grow, self.tokens.numel())
        self.calls.append((n, self.have))
        return self.tokens[:self.have].clone(), self.have == self.tokens.numel()


def test_live_token_source_yields_the_chunks_of_the_finished_sequence():
    """Whatever the pace at which tokens arrive (1, 7, 100, 120, 1000 at a time), the chunks equal those cut from the finished sequence,
    for lengths around every boundary of the schedule (hop + overlap = 120, 220, 320 ...), and the first chunk is cut as soon as
    120 tokens exist -- not when the sequence is complete."""
    cfg = SynthConfig()
    c = StreamConsts.for_config(cfg)
    for n in (1, 80, 119, 120, 121, 219, 220, 221, 330, 600):
        toks = torch.arange(n)
        log = []
        fm, f0, src, voc = _stages(cfg, log)
        ref = list(stream_render(toks, c, fm, f0, src, voc))
        ref_log = list(log)
        for grow in (1, 7, 100, 120, 1000):
            log.clear()
            s = _GrowingTokens(toks, grow)
            it = stream_render(s, c, fm, f0, src, voc)
            first = next(it)
            if n >= 120:
                assert s.have < n or n <= 120 + grow, (n, grow, s.have)         # the first chunk did not wait for the end
                assert s.calls[0][0] == 120
            got = [first] + list(it)
            assert log == ref_log, (n, grow)
            assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref)), (n, grow)
