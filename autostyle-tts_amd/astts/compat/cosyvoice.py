"""``CosyVoice`` / ``load_wav`` call surface of the reference, served by the MI355X synthesis engine.

Mirrors the calls the reference scripts make (paths under /root/reference):
  CosyVoice(model_dir)                                              tts_with_rag.py:159, tts_with_style_and_timbre.py:74
  .inference_tts_with_st(tts_text, style_text, style_wav_16k, timbre_wav_16k, stream=False)
                                                                    tts_with_rag.py:195, tts_with_style_and_timbre.py:93
  .inference_zero_shot(tts_text, prompt_text, prompt_wav_16k, stream=False)     tts_with_rag.py:133, basic.py:15
  .inference_vc(source_wav_16k, prompt_wav_16k, stream=False)                   tts_with_rag.py:141
  load_wav(path, target_sr) -> FloatTensor[1, n]                                tts_with_rag.py:180-186
Each method is a *generator* that yields one ``{'tts_speech': FloatTensor[1, n_samples]}`` (CPU) per text
segment; work for segment i happens on ``next()`` (lazy, as upstream).  With ``stream=True`` a segment is yielded as several
consecutive chunks (2 s hops, mel / waveform cross-fades: ``astts.synth.stream``).

``inference_tts_with_st`` exists only in the authors' fork; its contract is the docstring at
tts_with_rag.py:151-156: step 1, the LM generates speech ("style") tokens conditioned on
(text, style-wav text, style-wav tokens + style speaker embedding); step 2, the flow decoder + vocoder
render those tokens conditioned on the TIMBRE wav (its tokens, mel and speaker embedding).

Weights: ``model_dir`` holding ``llm.pt`` / ``flow.pt`` / ``hift.pt`` is loaded as is.  No checkpoint
exists offline, so an absent directory falls back to seeded random-init weights at the configured
shapes and says so loudly (``self.random_init``): the audio is then noise-like by construction.
Sampling, CFM noise and source phases come from a ``torch.Generator`` seeded per instance.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Dict, Iterator, Optional

import torch

from .. import audio
from ..frontend import Frontend, PromptFeatures, text_normalize
from ..synth.config import SynthConfig


def load_wav(path: str, target_sr: int) -> torch.Tensor:
    return audio.load_wav(path, target_sr)


class _LmTokenStream:
    """The speech tokens of one segment WHILE the LM decodes them (``stream=True``).  Upstream runs ``llm.inference`` on a thread that
    appends to a list and lets ``tts()`` cut a chunk as soon as hop + look-ahead tokens are there; here the decode chain is issued in
    ranges of steps (``AcousticLM.decode_range`` -> astts_lm_decode_range) by a worker thread on its OWN HIP stream, each range
    followed by an event and an asynchronous copy of its tokens to pinned memory, so that chunk k is rendered on the caller's stream
    while the chain decodes the tokens of chunk k + 1.  ``wait(n)`` returns the tokens so far once n exist (or all there will be):
    the interface ``synth.stream.stream_render`` takes in place of a finished token tensor."""

    def __init__(self, lm, state, uniforms, min_len: int, max_len: int, eos_id: int, ranges, stream: "torch.cuda.Stream", fixed: bool):
        import threading

        self.lm, self.eos, self.max_len, self.fixed = lm, eos_id, max_len, fixed
        self.stream = stream
        dev = lm.device
        stream.wait_stream(torch.cuda.current_stream(dev))          # the prefill ran on the caller's stream
        for t in lm.prefill_tensors(state) + [uniforms]:
            t.record_stream(stream)
        with torch.cuda.stream(stream):
            self.ctx = lm.decode_begin(state, uniforms, ignore_eos=max_len if fixed else min_len)
        self.host = torch.empty((max_len,), dtype=torch.int32).pin_memory()
        self.ranges = list(ranges)                                   # [(s_begin, s_end)] covering [0, max_len)
        self.events = [torch.cuda.Event() for _ in self.ranges]
        self.issued = [threading.Event() for _ in self.ranges]
        self.have, self.finished, self.cancel, self.error = 0, False, False, None
        self._done_ranges = 0
        self._cv = threading.Condition()
        self.thread = threading.Thread(target=self._issue, daemon=True)
        self.thread.start()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _issue(self):
        try:
            with torch.cuda.device(self.lm.device), torch.cuda.stream(self.stream):
                for k, (b, e) in enumerate(self.ranges):
                    # ONE range ahead of what the renderer has consumed (a range is ~100 steps x 72 launches = tens of ms of queued
                    # work: plenty to hide the launch latency): an early EOS / close() then cancels promptly instead of waiting for
                    # every range of the 20x window, and a queue shared with the renderer delays a chunk by one range at most
                    with self._cv:
                        while not self.cancel and k > self._done_ranges + 1:
                            self._cv.wait()
                    if self.cancel:
                        break
                    self.lm.decode_range(self.ctx, e)
                    self.host[b:e].copy_(self.ctx["toks"][0, b:e], non_blocking=True)
                    self.events[k].record(self.stream)
                    self.issued[k].set()
        except BaseException as e:      # noqa: BLE001  (re-raised in wait())
            self.error = e
        finally:
            for ev in self.issued:
                ev.set()

    def wait(self, n: int):
        n = min(int(n), self.max_len)
        while self.have < n and not self.finished:
            k = self._done_ranges
            self.issued[k].wait()
            if self.error is not None:
                raise self.error
            self.events[k].synchronize()                             # this range's tokens are in pinned memory
            b, e = self.ranges[k]
            with self._cv:
                self._done_ranges += 1
                self._cv.notify_all()
            row = self.host[b:e]
            eos = (row >= self.eos).nonzero() if not self.fixed else torch.empty(0)
            if eos.numel():                                          # the segment ends here: later ranges are not needed
                self.have = max(b + int(eos[0]), 1)
                self.finished, self.cancel = True, True
            else:
                self.have = e
                if self._done_ranges == len(self.ranges):
                    self.finished = True
        return self.host[:self.have].clone(), self.finished

    def close(self):
        with self._cv:
            self.cancel = True
            self._cv.notify_all()
        self.thread.join()
        torch.cuda.current_stream(self.lm.device).wait_stream(self.stream)


class CosyVoice:
    def __init__(self, model_dir: str, config: Optional[SynthConfig] = None, seed: int = 0, device=None,
                 frontend: Optional[Frontend] = None, allow_random_init: Optional[bool] = None, **_kw):
        """``model_dir`` must hold ``llm.pt`` / ``flow.pt`` / ``hift.pt`` (the CosyVoice-300M state dicts), as it must for
        the reference (tts_with_rag.py:159).  Without them the constructor RAISES -- a synthesis run on random weights
        writes noise and must not look like a success -- unless the caller opts in: ``allow_random_init=True`` (the CLIs'
        ``--allow_random_init``) or ``ASTTS_ALLOW_RANDOM_INIT=1`` (tests, benchmarks: no checkpoint exists offline).
        The frontend (text tokenizer vocabulary, speech tokenizer, speaker network: upstream's ``*.tiktoken``,
        ``speech_tokenizer_v1.onnx``, ``campplus.onnx``) is wired from the same directory (``Frontend.from_model_dir``); LOADED
        synthesis weights next to a missing frontend file RAISE too (``allow_standin_frontend=True`` /
        ``ASTTS_ALLOW_STANDIN_FRONTEND=1`` to run on synthetic frontend networks anyway)."""
        from ..synth.model import SynthEngine
        from ..synth.weights import load_state_dicts, make_all

        engine = _kw.pop("engine", None)          # an existing SynthEngine (benchmarks: one engine for several surfaces)
        allow_standin_frontend = _kw.pop("allow_standin_frontend", None)
        self.model_dir = model_dir
        if config is None:
            cfg_file = os.path.join(model_dir, "astts.json")          # this build's plain-JSON model config (sample rate, max_positions,
            if os.path.exists(cfg_file):                              # up-rates, vocabulary sizes ...: SynthConfig.from_json)
                config = SynthConfig.from_json(cfg_file)
            else:
                config = SynthConfig.tiny() if os.environ.get("ASTTS_TINY_MODEL") == "1" else SynthConfig()
        self.cfg = config
        self.sample_rate = config.sample_rate
        have = all(os.path.exists(os.path.join(model_dir, f"{n}.pt")) for n in ("llm", "flow", "hift"))
        if engine is not None:
            state, self.random_init = None, bool(getattr(engine, "random_init", not have))
        elif have:
            state = load_state_dicts(model_dir, config)               # held to the config's key / shape manifest: one readable error
            self.random_init = False
        else:
            if allow_random_init is None:
                allow_random_init = os.environ.get("ASTTS_ALLOW_RANDOM_INIT") == "1"
            if not allow_random_init:
                raise FileNotFoundError(f"CosyVoice: no llm.pt/flow.pt/hift.pt under {model_dir!r}.  Pass allow_random_init=True "
                                        f"(--allow_random_init / ASTTS_ALLOW_RANDOM_INIT=1) to run on seeded random weights.")
            warnings.warn(f"CosyVoice: no llm.pt/flow.pt/hift.pt under {model_dir!r}; using seeded RANDOM-INIT weights "
                          f"at the configured shapes (explicitly allowed)", stacklevel=2)
            state = make_all(config, seed)
            self.random_init = True
        self.engine = engine if engine is not None else SynthEngine(state, config, device)    # raises without a GPU: no CPU fallback
        self.device = self.engine.device
        if frontend is None:
            # real weights are never paired silently with an untrained frontend (an engine handed over in memory counts as loaded
            # only when it says so: engine.random_init = False)
            frontend = Frontend.from_model_dir(model_dir, config, self.device, weights_loaded=not self.random_init,
                                               allow_standins=allow_standin_frontend, seed=seed)
        self.frontend = frontend
        self._gen = torch.Generator().manual_seed(seed)
        self.min_token_text_ratio, self.max_token_text_ratio = 2, 20
        # stream=True: render chunk k while the LM decodes the tokens of chunk k + 1 (False: one decode pass, then the chunks -- rounds 3-4,
        # kept as the second implementation the tests compare the chunks with)
        self.stream_lm_live = os.environ.get("ASTTS_STREAM_LM_LIVE", "1") != "0"
        # ragged batches: the whole render group in one vocoder pass (False: one pass per row)
        self.vocoder_batched = os.environ.get("ASTTS_VOCODER_BATCHED", "1") != "0"
        # ragged batches: rows per LM job.  32 = the decode-step kernels, whose rows do not depend on the batch width (a row's audio is then
        # the same in every batch, rank and schedule).  Throughput runs may set ``wide_lm`` and up to 256 rows: one chain of plain GEMMs
        # per job (the weights are read once per token for all rows) -- the same arithmetic in another summation order, so a row's tokens
        # may differ from its 32-row run at near-ties of the sampler.
        self.wide_lm = os.environ.get("ASTTS_WIDE_LM", "0") == "1"
        self.lm_rows = int(os.environ.get("ASTTS_LM_ROWS", "48" if self.wide_lm else "32"))     # (48: the best of the config-4 sweep)
        # measurement only (bench.py): HIP events around the LM decode of every job and the flow / vocoder passes of every render group
        self.collect_stage_times = False
        self._stage_events = []
        self._host_times = []

    # ------------------------------------------------------------------ one text segment
    @staticmethod
    def segment_generator(seed: int, segment: int) -> torch.Generator:
        """The random stream of text segment ``segment`` of an item synthesised under ``seed`` (drivers: one seed per input row, so
        that a row's draws do not depend on the batch, the rank or the rows before it)."""
        return torch.Generator().manual_seed((int(seed) * 1000003 + int(segment)) & ((1 << 62) - 1))

    def _draws(self, n_tok: int, n_mel_total: int, n_mel_gen: int, gen: Optional[torch.Generator] = None):
        """The stochastic inputs of one segment from ``gen`` (default: the instance generator): sampling uniforms, CFM start noise and
        source phases on the host (small); the source NOISE -- 256 x 9 floats per mel frame, 24 MB for a 30 s row -- on the device, from
        a Philox stream whose seed is the next draw of ``gen`` (as upstream draws it on the device; on the host it cost more than the
        synthesis of a ragged test-set shard: 6.5 GB of Gaussians for the 1 623 IEMOCAP sentences)."""
        cfg, g = self.cfg, gen or self._gen
        nh = cfg.nb_harmonics + 1
        u = torch.rand(max(n_tok, 1), 1, 2, generator=g)
        z = torch.randn(1, n_mel_total, cfg.mel, generator=g)
        phase0 = (torch.rand(1, nh, generator=g) * 2 - 1) * math.pi
        phase0[:, 0] = 0
        seed = int(torch.randint(0, 1 << 62, (1,), generator=g))
        gd = torch.Generator(device=self.device).manual_seed(seed)
        noise = torch.randn(1, n_mel_gen * cfg.upsample_total, nh, generator=gd, device=self.device)
        return u, z, phase0, noise

    def _lm_setup(self, text_ids: torch.Tensor, n_tts_text: int, lm_prompt: PromptFeatures, gen: Optional[torch.Generator], fixed_tokens: Optional[int]):
        """Prefix, decode window and sampling uniforms of one segment: EOS masked for the first 2x text tokens, capped at 20x (upstream
        ratios); ``fixed_tokens``: exactly that many tokens, EOS ignored (throughput / latency runs: SURVEY.md 7)."""
        dev, lm = self.device, self.engine.lm
        tlen = torch.tensor([text_ids.shape[1]], dtype=torch.int32, device=dev)
        pre = lm.prefix(text_ids.to(dev), tlen, lm_prompt.spk_embedding.to(dev), lm_prompt.speech_tokens.to(dev))
        if fixed_tokens is not None:
            want = int(fixed_tokens)
            max_len = self._cap_tokens(want, pre.shape[0])
            min_len = max_len
        else:
            min_len = self.min_token_text_ratio * n_tts_text
            want = self.max_token_text_ratio * n_tts_text
            max_len = max(self._cap_tokens(want, pre.shape[0]), min_len + 1)
        # a FIXED count of uniforms per segment (upstream's 20x window), whatever the position-table cap: the later draws of the
        # segment's stream (CFM noise, phases, source noise) then do not shift when a cap applies
        u = torch.rand(max(want, max_len), 1, 2, generator=gen or self._gen)[:max_len].to(dev)
        return pre, min_len, max_len, u

    def _lm_tokens(self, text_ids: torch.Tensor, n_tts_text: int, lm_prompt: PromptFeatures, gen: Optional[torch.Generator] = None,
                   fixed_tokens: Optional[int] = None) -> torch.Tensor:
        """The segment's speech tokens from ONE on-device decode pass (one synchronisation per segment)."""
        cfg, lm = self.cfg, self.engine.lm
        pre, min_len, max_len, u = self._lm_setup(text_ids, n_tts_text, lm_prompt, gen, fixed_tokens)
        toks = lm.decode(pre, max_len, u, ignore_eos=True if fixed_tokens is not None else min_len)[0].cpu()
        eos = (toks >= cfg.speech_vocab).nonzero() if fixed_tokens is None else torch.empty(0)
        n = int(eos[0]) if eos.numel() else max_len
        return toks[:max(n, 1)][None, :].to(torch.int32)

    def _lm_token_stream(self, text_ids: torch.Tensor, n_tts_text: int, lm_prompt: PromptFeatures, gen: Optional[torch.Generator] = None,
                         fixed_tokens: Optional[int] = None) -> _LmTokenStream:
        """The same tokens as ``_lm_tokens`` (same prefix, window, uniforms: bit-identical), handed over WHILE they are decoded: the
        decode chain runs hop by hop on a stream of its own (first range = hop + look-ahead = 120 tokens, then 100 at a time: the
        chunk boundaries of ``synth.stream``), so the first chunk can be rendered after ~120 decode steps instead of after EOS."""
        from ..synth.stream import StreamConsts
        cfg, lm = self.cfg, self.engine.lm
        pre, min_len, max_len, u = self._lm_setup(text_ids, n_tts_text, lm_prompt, gen, fixed_tokens)
        c = StreamConsts.for_config(cfg)
        state = lm.prefill(pre, max_len)
        edges, e, hop = [0], c.token_min_hop + c.token_overlap, c.token_min_hop
        while e < max_len:
            edges.append(e)
            hop = min(c.token_max_hop, int(hop * c.scale))
            e += hop
        edges.append(max_len)
        # the decode chain goes beside the stream the CALLER renders on: another hardware queue on another command-processor pipe,
        # probed against that very stream (ops.stream_beside; cached per stream)
        from .. import ops
        lm_stream = ops.stream_beside(torch.cuda.current_stream(self.device), device=self.device)
        return _LmTokenStream(lm, state, u, min_len, max_len, cfg.speech_vocab, list(zip(edges[:-1], edges[1:])), lm_stream,
                              fixed_tokens is not None)

    def _cap_tokens(self, want: int, prefix_len: int, what: str = "this segment") -> int:
        """The relative-position tables bound prefix + generated tokens (SynthConfig.max_positions).  Upstream's tables extend, so a
        capped request is a divergence: it is reported, never silent."""
        cap = self.cfg.max_positions - int(prefix_len) - 100
        if want > cap:
            warnings.warn(f"CosyVoice: {what} may generate up to {want} speech tokens (upstream's {self.max_token_text_ratio}x text-length "
                          f"limit) but the position tables (SynthConfig.max_positions = {self.cfg.max_positions}) leave room for {max(cap, 1)} "
                          f"after a {int(prefix_len)}-position prefix: output capped.  Build the engine with a larger max_positions.",
                          RuntimeWarning, stacklevel=3)
        return max(1, min(int(want), cap))

    def _render(self, tokens: torch.Tensor, flow_prompt: PromptFeatures, gen: Optional[torch.Generator] = None) -> torch.Tensor:
        cfg, dev, eng = self.cfg, self.device, self.engine
        n_gen = cfg.mel_frames_for_tokens(tokens.shape[1])
        tmp = flow_prompt.mel.shape[1]
        _, z, phase0, noise = self._draws(0, tmp + n_gen, n_gen, gen)
        all_tok = torch.cat([flow_prompt.speech_tokens.to(torch.int32), tokens], dim=1).to(dev)
        tl = torch.tensor([all_tok.shape[1]], dtype=torch.int32, device=dev)
        mel = eng.flow.decode(all_tok, tl, flow_prompt.mel.to(dev), flow_prompt.spk_embedding.to(dev), z.to(dev), tmp + n_gen)
        wav = eng.hift.forward(mel, phase0.to(dev), noise.to(dev))
        return wav.cpu()

    def _render_stream(self, tokens, flow_prompt: PromptFeatures, gen: Optional[torch.Generator] = None) -> Iterator[torch.Tensor]:
        """``stream=True``: the segment's tokens -- a finished tensor (``inference_vc``: the source's tokens) or the live
        ``_LmTokenStream`` of a decode still running on its own stream -- rendered hop by hop with upstream's overlaps and cross-fades
        (``astts.synth.stream``).  With the live source the first chunk is yielded after ~120 decode steps, not after EOS."""
        from ..synth.stream import StreamConsts, stream_render
        cfg, dev, eng = self.cfg, self.device, self.engine
        nh = cfg.nb_harmonics + 1
        try:
            ptok = flow_prompt.speech_tokens.to(torch.int32)
            pmel, spk = flow_prompt.mel.to(dev), flow_prompt.spk_embedding.to(dev)
            tmp = pmel.shape[1]
            consts = StreamConsts.for_config(cfg)
        except BaseException:
            if hasattr(tokens, "close"):          # (a live decode is already running on its own stream: stop it)
                tokens.close()
            raise
        g = gen or self._gen            # the item's own random stream when it has one (seed=): chunks then do not depend on what ran before

        def flow_mel(tok: torch.Tensor) -> torch.Tensor:
            n_gen = cfg.mel_frames_for_tokens(int(tok.numel()))
            z = torch.randn(1, tmp + n_gen, cfg.mel, generator=g)
            all_tok = torch.cat([ptok, tok.view(1, -1).to(torch.int32)], dim=1).to(dev)
            tl = torch.tensor([all_tok.shape[1]], dtype=torch.int32, device=dev)
            return eng.flow.decode(all_tok, tl, pmel, spk, z.to(dev), tmp + n_gen)

        def source(f0: torch.Tensor) -> torch.Tensor:
            phase0 = (torch.rand(1, nh, generator=g) * 2 - 1) * math.pi
            phase0[:, 0] = 0
            noise = torch.randn(1, f0.shape[1] * cfg.upsample_total, nh, generator=g)
            return eng.hift.source(f0, phase0.to(dev), noise.to(dev))

        toks_in = tokens if hasattr(tokens, "wait") else tokens.view(-1)
        try:
            for wav in stream_render(toks_in, consts, flow_mel, eng.hift.f0, source, eng.hift.decode):
                yield wav.cpu()
        finally:
            if hasattr(tokens, "close"):
                tokens.close()

    def _emit(self, tokens: torch.Tensor, flow_prompt: PromptFeatures, stream: bool, gen: Optional[torch.Generator] = None) -> Iterator[Dict[str, torch.Tensor]]:
        if stream:
            for wav in self._render_stream(tokens, flow_prompt, gen):
                yield {"tts_speech": wav}
        else:
            yield {"tts_speech": self._render(tokens, flow_prompt, gen)}

    # ------------------------------------------------------------------ ragged batches (many segments in one pass)
    def synthesize_batch(self, requests, max_batch: int = 32, bucket: bool = True, fixed_tokens=None, draws=None, forced=None):
        """``requests``: list of (text_ids [1, Tt] = prompt text + segment text, n_segment_text_tokens, lm_prompt,
        flow_prompt).  The segments go through left-padded LM batches (per-row EOS window, tokens truncated at each row's EOS),
        ragged flow-matching batches and the vocoder.  Returns one FloatTensor[1, n] per request, each what the one-at-a-time
        path produces for that segment up to sampling draws.

        Schedule (the reference loops one utterance at a time, tts_with_rag.py:172-197; rows are independent):
          * rows are sorted by length (``bucket``) and cut into render groups of ``max_batch`` rows and into LM jobs of ``self.lm_rows``
            rows (32; up to 256 with ``self.wide_lm``);
          * the LM jobs -- latency-bound launch chains that leave most of the chip idle -- run on up to THREE worker threads with
            their own streams (one per command-processor pipe), each decoding only as far as its own longest row; the workers take
            jobs from one list sorted by length: all but the last the LONGEST remaining one, the last the SHORTEST (the render stage
            then has work from the first moments, and all streams run dry together);
          * the calling thread renders (flow matching + ONE vocoder pass per group) every group as soon as its tokens are there, in
            the order the jobs complete, and copies the results to the host asynchronously: the render stage of one group overlaps
            the decode chains of the next.
        Every request draws from its OWN random stream (``draws``; by default one ``torch.Generator`` per request seeded from the
        instance generator in request order), so the result does not depend on the schedule.

        Explicit control of the stochastic parts (SURVEY.md 7 "hard parts": throughput runs need fixed-length decode, parity
        needs injectable randomness), all per request, in request order:
          ``fixed_tokens[i]``  decode exactly that many speech tokens (EOS ignored) instead of the 2x..20x text-length window;
          ``draws[i]``         {"u": [n, 2], "z": [1, Tm_total, mel], "phase0": [1, nh], "noise": [1, L, nh]} instead of draws
                               from the instance generator (``make_draws`` builds them from a seed), or a ``torch.Generator``
                               that the request's draws are taken from (``segment_generator``: what the one-at-a-time path
                               draws for the same segment under the same seed);
          ``forced[i]``        teacher forcing: int tokens [n] that replace the sampled ones.
        The tokens and mels of the last call stay in ``self.last_tokens`` / ``self.last_mels`` (CPU)."""
        import threading
        import time

        cfg, dev, eng = self.cfg, self.device, self.engine
        t_entry = time.perf_counter()
        n_req = len(requests)
        out = [None] * n_req
        self.last_tokens, self.last_mels = [None] * n_req, [None] * n_req
        if n_req == 0:
            return out
        if draws is None:
            draws = [torch.Generator().manual_seed(int(torch.randint(0, 1 << 62, (1,), generator=self._gen))) for _ in range(n_req)]
        # length bucketing: a group is padded to its longest row in every stage, so rows of similar length go
        # together; results return in request order
        want = [int(fixed_tokens[i]) if fixed_tokens is not None else self.max_token_text_ratio * requests[i][1] for i in range(n_req)]
        order = sorted(range(n_req), key=lambda i: -want[i]) if bucket else list(range(n_req))
        rgroups = [order[g0:g0 + max_batch] for g0 in range(0, n_req, max_batch)]
        # LM jobs: consecutive rows of the sorted order, ``self.lm_rows`` at a time (32: the decode-step kernels; up to 256 with
        # ``wide_lm``: ONE chain of plain GEMMs for all rows, AcousticLM.decode(wide=True)).  A job may span several render groups and a
        # render group several jobs: a group is rendered once every job that holds one of its rows is done.
        lm_rows = max(1, min(int(self.lm_rows), 256 if self.wide_lm else 32))
        jobs = [(None, order[c0:c0 + lm_rows]) for c0 in range(0, n_req, lm_rows)]                # (unused, request indices)
        group_of = {i: gi for gi, g in enumerate(rgroups) for i in g}

        def lm_stage(idxs):
            tj0 = time.perf_counter()
            grp = [requests[i] for i in idxs]
            b = len(grp)
            pre, ks = eng.lm.prefix_ragged([r[0].view(-1) for r in grp], torch.cat([r[2].spk_embedding for r in grp], 0),
                                           [r[2].speech_tokens.view(-1) for r in grp])
            if fixed_tokens is not None:
                max_len = [self._cap_tokens(int(fixed_tokens[i]), pre.shape[0], f"request {i}") for i in idxs]   # the position tables bound prefix + tokens
                min_len = list(max_len)
            else:
                min_len = [self.min_token_text_ratio * r[1] for r in grp]
                max_len = [max(self._cap_tokens(self.max_token_text_ratio * r[1], pre.shape[0], f"request {i}"), m + 1)
                           for i, r, m in zip(idxs, grp, min_len)]
            n_steps = max(max_len)
            u = torch.zeros(n_steps, b, 2)
            for j, i in enumerate(idxs):
                d = draws[i]
                # a generator hands out a FIXED count per request (its uncapped window): `_cap_tokens` sees the padded prefix of the
                # whole LM job, so a capped row's length depends on its job (warned) -- but the rest of its random stream must not
                n_draw = max(want[i], max_len[j])
                u[:max_len[j], j] = torch.rand(n_draw, 2, generator=d)[:max_len[j]] if isinstance(d, torch.Generator) else d["u"][:max_len[j]]
            ft = None
            if forced is not None:
                ft = torch.zeros(b, n_steps, dtype=torch.int32)
                for j, i in enumerate(idxs):
                    ft[j, :max_len[j]] = forced[i][:max_len[j]].to(torch.int32)
                ft = ft.to(dev)
            eos_min = torch.tensor(min_len, dtype=torch.int32, device=dev)
            ev0 = self._stage_mark()
            th0 = time.perf_counter()
            toks = eng.lm.decode(pre, n_steps, u.to(dev), ignore_eos=eos_min, key_start=ks, forced_tokens=ft, wide=self.wide_lm)
            self._stage_mark("lm", ev0)
            th1 = time.perf_counter()
            toks = toks.cpu()                                                      # one sync per job (this thread's stream)
            if self.collect_stage_times:
                self._host_times.append(("lm_job_prepare", th0 - tj0))
                self._host_times.append(("lm_job_enqueue", th1 - th0))
                self._host_times.append(("lm_job_wait", time.perf_counter() - th1))
            gen = []
            for j in range(b):
                row = toks[j, :max_len[j]]
                eos = (row >= cfg.speech_vocab).nonzero() if fixed_tokens is None else torch.empty(0)
                n = int(eos[0]) if eos.numel() else max_len[j]
                gen.append(row[:max(n, 1)].to(torch.int32))
            return gen

        copies_done = []

        def release_staging(block: bool):
            """Results leave their page-locked staging as soon as their copies have landed: the caller gets pageable tensors (a kept
            waveform must not pin its whole render group's block -- a config-4 pass would leave GBs of host memory page-locked), and
            the block goes back to torch's pinned allocator for the next group."""
            rest = []
            for ev, wavs_d, mels_d, idxs_g in copies_done:
                if block:
                    ev.synchronize()
                elif not ev.query():
                    rest.append((ev, wavs_d, mels_d, idxs_g))
                    continue
                for i in idxs_g:
                    out[i] = out[i].clone()
                    self.last_mels[i] = self.last_mels[i].clone()
            copies_done[:] = rest

        def render(idxs, gen_tokens):
            tr0 = time.perf_counter()
            release_staging(block=False)
            all_tok, pmels, zs, dr = [], [], [], []
            for i in idxs:
                fp = requests[i][3]
                n_gen = cfg.mel_frames_for_tokens(int(gen_tokens[i].numel()))
                tmp = int(fp.mel.shape[1])
                d = draws[i]
                if isinstance(d, torch.Generator):
                    _, z, phase0, noise = self._draws(0, tmp + n_gen, n_gen, d)
                else:
                    z, phase0, noise = d["z"][:, :tmp + n_gen], d["phase0"], d["noise"][:, :n_gen * cfg.upsample_total]
                all_tok.append(torch.cat([fp.speech_tokens.view(-1).to(torch.int32), gen_tokens[i]]))
                pmels.append(fp.mel[0])
                zs.append(z[0])
                dr.append((phase0, noise))
            ev0 = self._stage_mark()
            mels = eng.flow.decode_ragged(all_tok, pmels, torch.cat([requests[i][3].spk_embedding for i in idxs], 0), zs)
            ev0 = self._stage_mark("flow", ev0)
            if self.vocoder_batched and len(idxs) > 1:
                # ONE vocoder pass over the ragged group: every convolution / STFT / iSTFT reads a row as a sequence of its own length
                # (HiftVocoder.forward_ragged); each waveform equals the row vocoded alone (tests/test_synth_gpu.py)
                wavs = eng.hift.forward_ragged(mels, [dr[j][0] for j in range(len(idxs))], [dr[j][1].to(dev) for j in range(len(idxs))])
            else:                       # one pass per row (rounds 1-4: ~250 launches each; kept as the second implementation)
                wavs = [eng.hift.forward(mels[j][None], dr[j][0].to(dev), dr[j][1].to(dev)) for j in range(len(idxs))]
            self._stage_mark("vocoder", ev0)
            # results go to the host ASYNCHRONOUSLY (one pinned staging buffer per group, views handed out; one event per group, waited
            # for at the end of the call): a blocking copy per row kept this thread from preparing the next group while the GPU worked
            # on this one -- the render stream idled for the host between groups
            n_w = [int(w.numel()) for w in wavs]
            n_m = [int(m.numel()) for m in mels]
            stage_w = torch.empty(sum(n_w), dtype=torch.float32, pin_memory=True)
            stage_m = torch.empty(sum(n_m), dtype=torch.float32, pin_memory=True)
            ow = om = 0
            for j, i in enumerate(idxs):
                stage_w[ow:ow + n_w[j]].copy_(wavs[j].reshape(-1), non_blocking=True)
                stage_m[om:om + n_m[j]].copy_(mels[j].reshape(-1), non_blocking=True)
                out[i] = stage_w[ow:ow + n_w[j]].view(1, -1)
                self.last_tokens[i] = gen_tokens[i]
                self.last_mels[i] = stage_m[om:om + n_m[j]].view(mels[j].shape)
                ow += n_w[j]
                om += n_m[j]
            ev = torch.cuda.Event()
            ev.record()
            copies_done.append((ev, wavs, mels, list(idxs)))          # (the device tensors stay alive until their copies have landed)
            if self.collect_stage_times:
                self._host_times.append(("render_group_host", time.perf_counter() - tr0))

        if len(jobs) == 1:                      # one job: nothing to overlap
            toks1 = dict(zip(jobs[0][1], lm_stage(jobs[0][1])))
            for g in rgroups:
                render(g, toks1)
            release_staging(block=True)
            return out
        # Schedule of the LM jobs (round 5).  The jobs are taken from ONE list sorted by cost (decode steps of the job's longest row): all
        # workers but the last take the LONGEST remaining job, the last worker the SHORTEST.  Rounds 3-4 ran every worker longest-first
        # on a static assignment: the first tokens then arrive when the first 1 500-step job ends -- the render stream idled for the
        # first ~3 s of a config-4 pass and the decode streams idled at the end while it caught up (LM 12.9 s per stream, render 12 s,
        # wall 17.1 s).  With a short-first worker the render stage has groups to work on from the first 0.1 s, the long groups arrive
        # in between, and the two ends of the list meet in the middle: every stream runs out of work at about the same time.  Render
        # groups are rendered in the order they COMPLETE.  (A request's randomness is its own, so the schedule does not show in the
        # results: tests/test_cli_gpu.py::test_batch_surface_does_not_depend_on_its_schedule.)
        import queue
        cost = [max(want[i] for i in idxs) for _, idxs in jobs]
        nw = max(1, min(int(os.environ.get("ASTTS_RAGGED_LM_WORKERS", "3")), 3, len(jobs)))
        pool = sorted(range(len(jobs)), key=lambda j: (-cost[j], j))          # longest first
        pool_lock = threading.Lock()
        short_first = os.environ.get("ASTTS_RAGGED_SHORT_WORKER", "1") != "0"
        done_q: "queue.Queue" = queue.Queue()
        if getattr(self, "_lm_streams", None) is None:
            # one stream per command-processor pipe, found by measurement (launch chains whose queues share a pipe take turns at every
            # kernel boundary: synth/model.py, PipelinedSynth): render, then the decode workers
            from .. import ops
            firsts = [c[0] for c in ops.stream_pipe_classes(device=dev)]
            if len(firsts) >= 4:
                self._render_stream_b, self._lm_streams = firsts[0], firsts[1:4]
            elif len(firsts) == 3:
                self._render_stream_b, self._lm_streams = firsts[0], firsts[1:3] + [firsts[1]]
            else:
                two = ops.concurrent_streams(2, device=dev)
                self._render_stream_b, self._lm_streams = torch.cuda.current_stream(dev), two + [two[0]]
        cur = torch.cuda.current_stream(dev)
        rs = self._render_stream_b

        def take(w):
            with pool_lock:
                if not pool:
                    return None
                return pool.pop() if (short_first and nw > 1 and w == nw - 1) else pool.pop(0)

        def worker(w):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(self._lm_streams[w]):
                    while True:
                        j = take(w)
                        if j is None:
                            break
                        done_q.put((j, lm_stage(jobs[j][1]), None))
            except BaseException as e:          # noqa: BLE001  (re-raised on the calling thread)
                with pool_lock:
                    pool.clear()
                done_q.put((-1, None, e))

        threads = []
        for w in range(nw):
            self._lm_streams[w].wait_stream(cur)
            threads.append(threading.Thread(target=worker, args=(w,)))
            threads[-1].start()
        try:
            rs.wait_stream(cur)
            if os.environ.get("ASTTS_RAGGED_EXCLUSIVE") == "1":       # experiment: every LM job first, then every render group
                for th in threads:
                    th.join()
            need = [len(g) for g in rgroups]                   # rows of each render group whose tokens are still missing
            toks_all = {}
            left = len(jobs)
            with torch.cuda.stream(rs):
                while left:
                    j, gen, err = done_q.get()
                    if err is not None:
                        raise err
                    left -= 1
                    toks_all.update(zip(jobs[j][1], gen))
                    ready = []
                    for i in jobs[j][1]:
                        gi = group_of[i]
                        need[gi] -= 1
                        if need[gi] == 0:
                            ready.append(gi)
                    for gi in ready:
                        render(rgroups[gi], toks_all)
        finally:
            with pool_lock:
                pool.clear()                    # (an error on this thread: the workers stop after their current job)
            for th in threads:
                th.join()
            for st in list(self._lm_streams) + [rs]:
                cur.wait_stream(st)
        t_sync = time.perf_counter()
        release_staging(block=True)             # every waveform / mel has landed and left its staging block
        if self.collect_stage_times:
            self._host_times.append(("final_copy_wait", time.perf_counter() - t_sync))
            self._host_times.append(("synthesize_batch_wall", time.perf_counter() - t_entry))
        return out

    def _stage_mark(self, name: Optional[str] = None, since=None):
        """``collect_stage_times``: an event on the current stream; with ``name`` the span since ``since`` is booked under it."""
        if not self.collect_stage_times:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if name is not None and since is not None:
            self._stage_events.append((name, since, ev))          # list.append is atomic: LM workers and the render thread share it
        return ev

    def stage_seconds(self) -> Dict[str, float]:
        """Stream-seconds per stage since the last call (``collect_stage_times``): the LM jobs run on two streams beside the render
        stream, so the three figures overlap in wall time and may sum to more than it."""
        torch.cuda.synchronize(self.device)
        out: Dict[str, float] = {}
        for name, a, b in self._stage_events:
            out[name] = out.get(name, 0.0) + a.elapsed_time(b) / 1e3
        for name, dt in self._host_times:                      # host-side wall seconds (thread-seconds for the per-job items)
            out["host_" + name] = out.get("host_" + name, 0.0) + dt
        self._stage_events, self._host_times = [], []
        return out

    def make_draws(self, n_tokens: int, prompt_mel_frames: int, seed: int):
        """The stochastic inputs of one segment (sampling uniforms, CFM start noise, source phases / noise) from a seed."""
        cfg = self.cfg
        g = torch.Generator().manual_seed(seed)
        nh = cfg.nb_harmonics + 1
        n_gen = cfg.mel_frames_for_tokens(n_tokens)
        phase0 = (torch.rand(1, nh, generator=g) * 2 - 1) * math.pi
        phase0[:, 0] = 0
        return {"u": torch.rand(n_tokens, 2, generator=g), "z": torch.randn(1, prompt_mel_frames + n_gen, cfg.mel, generator=g),
                "phase0": phase0, "noise": torch.randn(1, n_gen * cfg.upsample_total, nh, generator=g)}

    def inference_tts_with_st_batch(self, items, max_batch: int = 32, split: bool = True, seeds=None, **controls):
        """Batched form of inference_tts_with_st for drivers that know all their work up front
        (tts_with_rag.py:172 loops 64 rows one by one).  ``items``: list of (tts_text, style_text, style_wav_16k,
        timbre_wav_16k) -> list (per item) of lists (per text segment) of {'tts_speech': FloatTensor[1, n]}.
        ``controls`` (fixed_tokens / draws / forced, per text segment; ``split=False`` keeps one segment per item so that they
        line up with the items) go to ``synthesize_batch``; the prompts of an item are featurised once per distinct tensor.
        ``seeds``: one integer per item -- segment k of item i then draws from ``segment_generator(seeds[i], k)``, exactly what
        ``inference_tts_with_st(..., seed=seeds[i])`` draws for it: an item's randomness is its own, whatever batch or rank it is in."""
        import time
        t_build = time.perf_counter()
        fe = self.frontend
        reqs, owner = [], []
        gens = [] if seeds is not None else None
        if seeds is not None and "draws" in controls:
            raise ValueError("inference_tts_with_st_batch: pass seeds or draws, not both")
        cache = {}
        distinct = {}
        for (_t, _s, style_wav, timbre_wav) in items:        # every distinct prompt tensor once, equal lengths as one GPU batch
            for w in (style_wav, timbre_wav):
                distinct.setdefault(id(w), w)
        if hasattr(fe, "prompts"):
            for k_, f_ in zip(distinct, fe.prompts(list(distinct.values()))):
                cache[k_] = f_

        def prompt(w):
            k = id(w)
            if k not in cache:
                cache[k] = fe.prompt(w)
            return cache[k]

        for k, (tts_text, style_text, style_wav, timbre_wav) in enumerate(items):
            style, timbre = prompt(style_wav), prompt(timbre_wav)
            style_ids = fe.text_ids(style_text)
            for n_seg, seg in enumerate(text_normalize(tts_text, fe.tokenizer, split=split)):
                seg_ids = fe.text_ids(seg)
                reqs.append((torch.cat([style_ids, seg_ids], dim=1), seg_ids.shape[1], style, timbre))
                owner.append(k)
                if gens is not None:
                    gens.append(self.segment_generator(seeds[k], n_seg))
        if gens is not None:
            controls["draws"] = gens
        if isinstance(controls.get("fixed_tokens"), int):        # one length for every text segment (throughput runs: SURVEY.md 7)
            controls["fixed_tokens"] = [controls["fixed_tokens"]] * len(reqs)
        if self.collect_stage_times:
            import time
            self._host_times.append(("build_requests", time.perf_counter() - t_build))
        wavs = self.synthesize_batch(reqs, max_batch, **controls)
        out = [[] for _ in items]
        for k, w in zip(owner, wavs):
            out[k].append({"tts_speech": w})
        return out

    # ------------------------------------------------------------------ public generators
    def inference_tts_with_st(self, tts_text: str, style_text: str, style_wav_16k: torch.Tensor,
                              timbre_wav_16k: torch.Tensor, stream: bool = False, seed: Optional[int] = None,
                              fixed_tokens: Optional[int] = None) -> Iterator[Dict[str, torch.Tensor]]:
        """``seed`` (an addition to the reference's signature): the item's own random stream (``segment_generator``) instead of
        the instance generator.  ``fixed_tokens`` (likewise): every segment decodes exactly that many speech tokens, EOS ignored."""
        fe = self.frontend
        style = fe.prompt(style_wav_16k)
        timbre = fe.prompt(timbre_wav_16k)
        style_ids = fe.text_ids(style_text)
        for n_seg, seg in enumerate(text_normalize(tts_text, fe.tokenizer, split=True)):
            seg_ids = fe.text_ids(seg)
            text_ids = torch.cat([style_ids, seg_ids], dim=1)
            gen = None if seed is None else self.segment_generator(seed, n_seg)
            if stream and self.stream_lm_live:      # step 1 WHILE step 2 renders the chunks it has the tokens for
                toks = self._lm_token_stream(text_ids, seg_ids.shape[1], style, gen, fixed_tokens)
            else:
                toks = self._lm_tokens(text_ids, seg_ids.shape[1], style, gen, fixed_tokens)  # step 1: style tokens
            yield from self._emit(toks, timbre, stream, gen)                 # step 2: render with the timbre

    def inference_zero_shot(self, tts_text: str, prompt_text: str, prompt_wav_16k: torch.Tensor,
                            stream: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
        fe = self.frontend
        prompt = fe.prompt(prompt_wav_16k)
        prompt_ids = fe.text_ids(prompt_text)
        for seg in text_normalize(tts_text, fe.tokenizer, split=True):
            seg_ids = fe.text_ids(seg)
            lm_in = (torch.cat([prompt_ids, seg_ids], dim=1), seg_ids.shape[1], prompt)
            toks = self._lm_token_stream(*lm_in) if stream and self.stream_lm_live else self._lm_tokens(*lm_in)
            yield from self._emit(toks, prompt, stream)

    def inference_vc(self, source_wav_16k: torch.Tensor, prompt_wav_16k: torch.Tensor,
                     stream: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
        fe = self.frontend
        src = fe.prompt(source_wav_16k)
        prompt = fe.prompt(prompt_wav_16k)
        yield from self._emit(src.speech_tokens, prompt, stream)               # no LM: source tokens, prompt timbre
