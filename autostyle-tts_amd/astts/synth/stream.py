"""Chunked ("stream=True") rendering of one utterance's speech tokens: the reference's call sites pass ``stream=False``
(tts_with_rag.py:195, tts_with_style_and_timbre.py:93, basic.py:15) but the methods take the flag; this
restates what upstream CosyVoice's ``CosyVoiceModel.tts(stream=True)`` / ``token2wav`` do with it [EXT: cosyvoice/cli/model.py,
CosyVoice-300M generation]:

* tokens are rendered in hops of ``token_min_hop_len`` = 2 s (100 tokens; grows by ``stream_scale_factor`` up to 4 s), each hop
  together with ``token_overlap_len`` = 20 look-ahead tokens, the flow decoder conditioned on the prompt every time;
* consecutive mels are cross-faded over ``mel_overlap_len`` = 34 frames with a Hamming window (the last 34 frames of a
  non-final chunk are withheld and blended into the head of the next one);
* the vocoder sees the last ``mel_cache_len`` = 20 frames of the previous chunk again, its harmonic source for those frames is
  the cached one (no phase glitch), and the first 20 x 256 samples are cross-faded with the previous chunk's withheld tail.

The stage functions are passed in, so the same state machine runs over the HIP engine (product: ``compat.cosyvoice``) and, in
tests only, over ``oracle/`` -- this module imports neither."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Iterator, Optional

import torch


@dataclass
class StreamConsts:
    token_min_hop: int
    token_max_hop: int
    token_overlap: int
    mel_overlap: int
    mel_cache: int
    source_cache: int
    scale: float = 1.0

    @staticmethod
    def for_config(cfg) -> "StreamConsts":
        return StreamConsts(token_min_hop=2 * cfg.token_rate, token_max_hop=4 * cfg.token_rate, token_overlap=20,
                            mel_overlap=int(20 / cfg.token_rate * cfg.sample_rate / cfg.hop), mel_cache=20,
                            source_cache=20 * cfg.hop)


def hamming(n: int) -> torch.Tensor:
    """numpy.hamming(n) (symmetric), fp32."""
    if n == 1:
        return torch.ones(1)
    k = torch.arange(n, dtype=torch.float64)
    return (0.54 - 0.46 * torch.cos(2.0 * math.pi * k / (n - 1))).float()


def _fade(new: torch.Tensor, old_tail: torch.Tensor, window: torch.Tensor) -> torch.Tensor:
    """upstream ``fade_in_out`` along dim 1: the head of ``new`` rises with the first half of the window while the tail of the
    previous chunk falls with the second half."""
    n = window.shape[0] // 2
    n = min(n, new.shape[1], old_tail.shape[1])
    if n == 0:
        return new
    w = window.to(new.device)
    half = window.shape[0] // 2
    shape = [1, n] + [1] * (new.dim() - 2)
    new = new.clone()
    new[:, :n] = new[:, :n] * w[:n].view(shape) + old_tail[:, -n:] * w[half:half + n].view(shape)
    return new


def stream_render(tokens: torch.Tensor, consts: StreamConsts, flow_mel: Callable[[torch.Tensor], torch.Tensor],
                  f0: Callable[[torch.Tensor], torch.Tensor], source: Callable[[torch.Tensor], torch.Tensor],
                  vocode: Callable[[torch.Tensor, torch.Tensor], torch.Tensor]) -> Iterator[torch.Tensor]:
    """``tokens``: int [n].  ``flow_mel(tok [n_i]) -> mel [1, Tm_i, 80]`` (prompt conditioning and noise inside),
    ``f0(mel) -> [1, Tm]``, ``source(f0) -> [1, Tm * hop]`` (phases / noise inside), ``vocode(mel, source) -> wav [1, Tm * hop]``.
    Yields the utterance as consecutive waveform chunks ``[1, n_i]`` (on whatever device the stage functions use)."""
    c = consts
    mel_window, speech_window = hamming(2 * c.mel_overlap), hamming(2 * c.source_cache)
    mel_overlap: Optional[torch.Tensor] = None
    cache = None

    def token2wav(tok: torch.Tensor, finalize: bool) -> torch.Tensor:
        nonlocal mel_overlap, cache
        mel = flow_mel(tok)
        if mel_overlap is not None:
            mel = _fade(mel, mel_overlap, mel_window)
        if cache is not None:
            mel = torch.cat([cache["mel"], mel], dim=1)
        if not finalize:
            mel_overlap = mel[:, -c.mel_overlap:]
            mel = mel[:, :-c.mel_overlap]
        src = source(f0(mel))
        if cache is not None:
            n = min(cache["source"].shape[1], src.shape[1])
            src = src.clone()
            src[:, :n] = cache["source"][:, :n]
        wav = vocode(mel, src)
        if cache is not None:
            wav = _fade(wav, cache["speech"], speech_window)
        if not finalize:
            cache = {"mel": mel[:, -c.mel_cache:], "source": src[:, -c.source_cache:], "speech": wav[:, -c.source_cache:]}
            wav = wav[:, :-c.source_cache]
        return wav

    # ``tokens`` is the finished token sequence (a tensor), or a token SOURCE with ``wait(n) -> (tokens so far [>= n, or all there will
    # be], finished)`` -- the LM still decoding on its own stream (upstream: the LM thread appends to a list that tts() polls).  The
    # schedule is the same either way: a non-final chunk is cut as soon as hop + overlap tokens exist beyond the current offset.
    src = tokens if hasattr(tokens, "wait") else _FixedTokens(tokens)
    off, hop = 0, c.token_min_hop
    while True:
        have, finished = src.wait(off + hop + c.token_overlap)
        if have.numel() - off >= hop + c.token_overlap:
            yield token2wav(have[off:off + hop + c.token_overlap], False)
            off += hop
            hop = min(c.token_max_hop, int(hop * c.scale))
            continue
        assert finished, "a token source returned fewer tokens than asked for without being finished"
        yield token2wav(have[off:], True)
        return


class _FixedTokens:
    def __init__(self, tokens: torch.Tensor):
        self.tokens = tokens.reshape(-1)

    def wait(self, n: int):
        return self.tokens, True
