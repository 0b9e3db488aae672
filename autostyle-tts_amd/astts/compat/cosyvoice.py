"""``CosyVoice`` / ``load_wav`` call surface of the reference, served by the MI355X synthesis engine.

Mirrors the calls the reference scripts make (paths under /root/reference):
  CosyVoice(model_dir)                                              tts_with_rag.py:159, tts_with_style_and_timbre.py:74
  .inference_tts_with_st(tts_text, style_text, style_wav_16k, timbre_wav_16k, stream=False)
                                                                    tts_with_rag.py:195, tts_with_style_and_timbre.py:93
  .inference_zero_shot(tts_text, prompt_text, prompt_wav_16k, stream=False)     tts_with_rag.py:133, basic.py:15
  .inference_vc(source_wav_16k, prompt_wav_16k, stream=False)                   tts_with_rag.py:141
  load_wav(path, target_sr) -> FloatTensor[1, n]                                tts_with_rag.py:180-186
Each method is a *generator* that yields one ``{'tts_speech': FloatTensor[1, n_samples]}`` (CPU) per text
segment; work for segment i happens on ``next()`` (lazy, as upstream).

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


class CosyVoice:
    def __init__(self, model_dir: str, config: Optional[SynthConfig] = None, seed: int = 0, device=None,
                 frontend: Optional[Frontend] = None, **_kw):
        from ..synth.model import SynthEngine
        from ..synth.weights import load_state_dicts, make_all

        self.model_dir = model_dir
        if config is None:
            config = SynthConfig.tiny() if os.environ.get("ASTTS_TINY_MODEL") == "1" else SynthConfig()
        self.cfg = config
        self.sample_rate = config.sample_rate
        have = all(os.path.exists(os.path.join(model_dir, f"{n}.pt")) for n in ("llm", "flow", "hift"))
        if have:
            state = load_state_dicts(model_dir)
            self.random_init = False
        else:
            warnings.warn(f"CosyVoice: no llm.pt/flow.pt/hift.pt under {model_dir!r}; using seeded RANDOM-INIT weights "
                          f"at the configured shapes (no checkpoint is available offline)", stacklevel=2)
            state = make_all(config, seed)
            self.random_init = True
        self.engine = SynthEngine(state, config, device)          # raises without a GPU: no CPU fallback
        self.device = self.engine.device
        self.frontend = frontend or Frontend(config)
        self._gen = torch.Generator().manual_seed(seed)
        self.min_token_text_ratio, self.max_token_text_ratio = 2, 20

    # ------------------------------------------------------------------ one text segment
    def _draws(self, n_tok: int, n_mel_total: int, n_mel_gen: int):
        cfg, g = self.cfg, self._gen
        nh = cfg.nb_harmonics + 1
        u = torch.rand(max(n_tok, 1), 1, 2, generator=g)
        z = torch.randn(1, n_mel_total, cfg.mel, generator=g)
        phase0 = (torch.rand(1, nh, generator=g) * 2 - 1) * math.pi
        phase0[:, 0] = 0
        noise = torch.randn(1, n_mel_gen * cfg.upsample_total, nh, generator=g)
        return u, z, phase0, noise

    def _lm_tokens(self, text_ids: torch.Tensor, n_tts_text: int, lm_prompt: PromptFeatures) -> torch.Tensor:
        """LM decode with EOS: masked for the first 2x text tokens, capped at 20x (upstream ratios)."""
        cfg, dev, lm = self.cfg, self.device, self.engine.lm
        tlen = torch.tensor([text_ids.shape[1]], dtype=torch.int32, device=dev)
        pre = lm.prefix(text_ids.to(dev), tlen, lm_prompt.spk_embedding.to(dev), lm_prompt.speech_tokens.to(dev))
        min_len = self.min_token_text_ratio * n_tts_text
        max_len = min(self.max_token_text_ratio * n_tts_text, cfg.max_positions - pre.shape[0] - 100)
        max_len = max(max_len, min_len + 1)
        u = torch.rand(max_len, 1, 2, generator=self._gen).to(dev)
        toks = lm.decode(pre, max_len, u, ignore_eos=min_len)[0].cpu()       # one sync per segment
        eos = (toks >= cfg.speech_vocab).nonzero()
        n = int(eos[0]) if eos.numel() else max_len
        return toks[:max(n, 1)][None, :].to(torch.int32)

    def _render(self, tokens: torch.Tensor, flow_prompt: PromptFeatures) -> torch.Tensor:
        cfg, dev, eng = self.cfg, self.device, self.engine
        n_gen = cfg.mel_frames_for_tokens(tokens.shape[1])
        tmp = flow_prompt.mel.shape[1]
        _, z, phase0, noise = self._draws(0, tmp + n_gen, n_gen)
        all_tok = torch.cat([flow_prompt.speech_tokens.to(torch.int32), tokens], dim=1).to(dev)
        tl = torch.tensor([all_tok.shape[1]], dtype=torch.int32, device=dev)
        mel = eng.flow.decode(all_tok, tl, flow_prompt.mel.to(dev), flow_prompt.spk_embedding.to(dev), z.to(dev), tmp + n_gen)
        wav = eng.hift.forward(mel, phase0.to(dev), noise.to(dev))
        return wav.cpu()

    # ------------------------------------------------------------------ public generators
    def inference_tts_with_st(self, tts_text: str, style_text: str, style_wav_16k: torch.Tensor,
                              timbre_wav_16k: torch.Tensor, stream: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
        fe = self.frontend
        style = fe.prompt(style_wav_16k)
        timbre = fe.prompt(timbre_wav_16k)
        style_ids = fe.text_ids(style_text)
        for seg in text_normalize(tts_text, fe.tokenizer, split=True):
            seg_ids = fe.text_ids(seg)
            text_ids = torch.cat([style_ids, seg_ids], dim=1)
            toks = self._lm_tokens(text_ids, seg_ids.shape[1], style)       # step 1: style tokens
            yield {"tts_speech": self._render(toks, timbre)}                 # step 2: render with the timbre

    def inference_zero_shot(self, tts_text: str, prompt_text: str, prompt_wav_16k: torch.Tensor,
                            stream: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
        fe = self.frontend
        prompt = fe.prompt(prompt_wav_16k)
        prompt_ids = fe.text_ids(prompt_text)
        for seg in text_normalize(tts_text, fe.tokenizer, split=True):
            seg_ids = fe.text_ids(seg)
            toks = self._lm_tokens(torch.cat([prompt_ids, seg_ids], dim=1), seg_ids.shape[1], prompt)
            yield {"tts_speech": self._render(toks, prompt)}

    def inference_vc(self, source_wav_16k: torch.Tensor, prompt_wav_16k: torch.Tensor,
                     stream: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
        fe = self.frontend
        src = fe.prompt(source_wav_16k)
        prompt = fe.prompt(prompt_wav_16k)
        yield {"tts_speech": self._render(src.speech_tokens, prompt)}          # no LM: source tokens, prompt timbre
