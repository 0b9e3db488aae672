"""Frontend of the CosyVoice call surface: text -> token ids, prompt wav -> speech tokens / speaker
embedding / mel.  Outside the measured GPU path.

What the reference uses here is NOT available offline and cannot be re-created from first
principles: the multilingual BPE vocabulary (tiktoken file), ``speech_tokenizer_v1.onnx`` and
``campplus.onnx`` (trained networks run through onnxruntime).  So every learned component is a
PLUGGABLE interface with a deterministic, clearly labelled stand-in:

  ByteTokenizer          UTF-8 bytes -> ids           (stand-in for the 51 866-entry BPE)
  EnergyVQSpeechTokenizer 20 ms log-mel frames -> ids  (stand-in for speech_tokenizer_v1.onnx, 50 Hz)
  StatsSpeakerEmbedder   pooled fbank stats -> 192-d   (stand-in for campplus.onnx)

The stand-ins produce inputs of exactly the shapes/rates/dtypes the real models produce, so the
GPU path (which only sees ids, embeddings and mels) is exercised identically.  Precomputed real
features can be injected instead through ``Frontend(features=...)``.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import audio
from .synth.config import SynthConfig


class ByteTokenizer:
    """Deterministic stand-in tokenizer: UTF-8 bytes offset into the id space (ids < vocab)."""

    def __init__(self, vocab: int):
        self.vocab = vocab

    def encode(self, text: str) -> List[int]:
        return [(b + 3) % self.vocab for b in text.encode("utf-8")]


class EnergyVQSpeechTokenizer:
    """Deterministic stand-in for the ONNX speech tokenizer: 16 kHz wav -> 50 tokens/s.
    Each 20 ms hop gets the arg-max of a fixed random projection of its 128-bin log-mel frame."""

    def __init__(self, n_codes: int, seed: int = 1234):
        g = torch.Generator().manual_seed(seed)
        self.proj = torch.randn(128, n_codes, generator=g)
        self.n_codes = n_codes
        self.device = None          # set by Frontend: the log-mel then runs as the HIP kernel (audio.mel_spectrogram)

    def __call__(self, wav16k: torch.Tensor) -> torch.Tensor:
        n = wav16k.shape[-1]
        if n > 30 * 16000:
            raise ValueError("do not support extract speech token for audio longer than 30s")  # upstream assert
        w = wav16k.to(self.device) if self.device is not None else wav16k
        mel = audio.mel_spectrogram(w, sr=16000, n_fft=400, hop=320, win=400, n_mels=128, fmin=0.0, fmax=8000.0).cpu()
        return torch.argmax(mel[0] @ self.proj, dim=-1).to(torch.int32)[None, :]            # [1, T] @ 50 Hz


class StatsSpeakerEmbedder:
    """Deterministic stand-in for the speaker-embedding network: pooled statistics of its real INPUT -- the 80-bin Kaldi fbank of the
    16 kHz prompt (``audio.kaldi_fbank``: what upstream's frontend hands campplus.onnx, before the mean over time is removed) -> 192-d."""

    def __init__(self, dim: int, seed: int = 4321):
        g = torch.Generator().manual_seed(seed)
        self.proj = torch.randn(160, dim, generator=g) / np.sqrt(160.0)
        self.device = None

    def __call__(self, wav16k: torch.Tensor) -> torch.Tensor:
        w = wav16k.to(self.device) if self.device is not None else wav16k
        mel = audio.kaldi_fbank(w, n_mels=80)[0].cpu()            # [frames, 80]; upstream subtracts the mean over time before the network
        stats = torch.cat([mel.mean(dim=0) - mel.mean(), (mel - mel.mean(dim=0, keepdim=True)).std(dim=0)])
        return (stats @ self.proj)[None, :]                                                 # [1, dim]


_SPLIT = re.compile(r"(?<=[。！？!?；;.…\n])\s*")


def text_normalize(text: str, tokenizer, split: bool = True, token_max_n: int = 80, token_min_n: int = 60,
                   merge_len: int = 20) -> List[str]:
    """cosyvoice frontend.text_normalize(split=True): strip, cut at sentence punctuation into segments of at most
    ``token_max_n`` tokens, merge trailing fragments shorter than ``merge_len`` tokens into their predecessor.
    (No inflection / number verbalisation: WeTextProcessing is not available offline.)"""
    text = text.strip()
    if not split:
        return [text]
    pieces = [p for p in _SPLIT.split(text) if p.strip()]
    segs: List[str] = []
    cur = ""
    for p in pieces:
        if cur and len(tokenizer.encode(cur + p)) > token_max_n and len(tokenizer.encode(cur)) >= min(token_min_n, token_max_n):
            segs.append(cur)
            cur = p
        elif cur and len(tokenizer.encode(cur + p)) > token_max_n:
            segs.append(cur)
            cur = p
        else:
            cur = (cur + " " + p).strip() if cur else p
    if cur:
        if segs and len(tokenizer.encode(cur)) < merge_len:
            segs[-1] = segs[-1] + " " + cur
        else:
            segs.append(cur)
    return segs or [text]


@dataclass
class PromptFeatures:
    speech_tokens: torch.Tensor   # int32 [1, Tp]   (50 Hz)
    spk_embedding: torch.Tensor   # fp32 [1, spk_dim]
    mel: torch.Tensor             # fp32 [1, Tm_p, 80] at the model sample rate


class Frontend:
    def __init__(self, cfg: SynthConfig, tokenizer=None, speech_tokenizer: Optional[Callable] = None,
                 speaker_embedder: Optional[Callable] = None, features: Optional[Dict[str, PromptFeatures]] = None, device=None):
        self.cfg = cfg
        self.device = device            # a CUDA device: the resampler and the prompt mel run as HIP kernels (astts/audio.py)
        self.tokenizer = tokenizer or ByteTokenizer(cfg.text_vocab)
        self.speech_tokenizer = speech_tokenizer or EnergyVQSpeechTokenizer(cfg.speech_vocab)
        self.speaker_embedder = speaker_embedder or StatsSpeakerEmbedder(cfg.spk_dim)
        if device is not None:
            for part in (self.speech_tokenizer, self.speaker_embedder):
                if hasattr(part, "device") and part.device is None:
                    part.device = device
        self.features = features or {}

    def text_ids(self, text: str) -> torch.Tensor:
        ids = self.tokenizer.encode(text) or [0]
        return torch.tensor([ids], dtype=torch.int64)

    def prompt(self, wav16k: torch.Tensor, key: Optional[str] = None) -> PromptFeatures:
        """16 kHz mono prompt -> speech tokens, speaker embedding and the mel at the model rate; the mel and the
        token sequence are trimmed to the same duration (token_len = min(mel_len / 2, token_len), upstream)."""
        if key is not None and key in self.features:
            return self.features[key]
        cfg = self.cfg
        tok = self.speech_tokenizer(wav16k)
        emb = self.speaker_embedder(wav16k)
        src = wav16k.to(self.device) if self.device is not None else wav16k
        wav_sr = audio.resample(src, 16000, cfg.sample_rate)
        mel = audio.mel_spectrogram(wav_sr, sr=cfg.sample_rate, n_fft=1024, hop=cfg.hop, win=1024, n_mels=cfg.mel,
                                    fmin=0.0, fmax=8000.0).cpu()
        n_tok = min(tok.shape[1], int(mel.shape[1] * cfg.token_rate * cfg.hop / cfg.sample_rate))
        n_mel = cfg.mel_frames_for_tokens(n_tok)
        return PromptFeatures(tok[:, :n_tok].contiguous(), emb, mel[:, :n_mel].contiguous())
