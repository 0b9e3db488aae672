"""Frontend of the CosyVoice call surface: text -> token ids, prompt wav -> speech tokens / speaker
embedding / mel.  Outside the measured GPU path.

What the reference uses here: the multilingual BPE vocabulary (a tiktoken file), ``speech_tokenizer_v1.onnx`` and
``campplus.onnx`` (trained networks run through onnxruntime).  Round 6: the two NETWORKS run on the GPU
(astts/frontend_nets.py: a Whisper-style encoder + codebook search, CAM++), on the weights found in ``model_dir``
(``speech_tokenizer_v1.{pt,onnx}`` / ``campplus.{pt,onnx}``) or -- like every other stage when no checkpoint exists offline -- on
seeded synthetic weights at the published shapes; the BPE needs its vocabulary FILE (``model_dir/*.tiktoken``; astts/bpe.py).
``Frontend.from_model_dir`` wires them and REFUSES to pair loaded synthesis weights with anything that is not the real thing
(``allow_standins`` / ASTTS_ALLOW_STANDIN_FRONTEND=1 to override): a real LM fed byte ids and tokens from an untrained encoder
produces garbage audio with status OK.

CPU-only stand-ins (no GPU, no weights; used by host-side tests and as explicit plug-ins):
  ByteTokenizer          UTF-8 bytes -> ids           (stand-in for the 51 866-entry BPE)
  EnergyVQSpeechTokenizer 20 ms log-mel frames -> ids  (stand-in for the speech tokenizer, 50 Hz)
  StatsSpeakerEmbedder   pooled fbank stats -> 192-d   (stand-in for the speaker network)
All produce inputs of exactly the shapes / rates / dtypes the real models produce.  Precomputed real features can be injected
instead through ``Frontend(features=...)``.
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
    standin = True

    def __init__(self, vocab: int):
        self.vocab = vocab

    def encode(self, text: str) -> List[int]:
        return [(b + 3) % self.vocab for b in text.encode("utf-8")]


class EnergyVQSpeechTokenizer:
    """Deterministic stand-in for the ONNX speech tokenizer: 16 kHz wav -> 50 tokens/s.
    Each 20 ms hop gets the arg-max of a fixed random projection of its 128-bin log-mel frame."""
    standin = True

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
    standin = True

    def __init__(self, dim: int, seed: int = 4321):
        g = torch.Generator().manual_seed(seed)
        self.proj = torch.randn(160, dim, generator=g) / np.sqrt(160.0)
        self.device = None

    def __call__(self, wav16k: torch.Tensor) -> torch.Tensor:
        w = wav16k.to(self.device) if self.device is not None else wav16k
        mel = audio.kaldi_fbank(w, n_mels=80)[0].cpu()            # [frames, 80]; upstream subtracts the mean over time before the network
        stats = torch.cat([mel.mean(dim=0) - mel.mean(), (mel - mel.mean(dim=0, keepdim=True)).std(dim=0)])
        return (stats @ self.proj)[None, :]                                                 # [1, dim]


# ------------------------------------------------------------------------------------------ text normalisation
# upstream cosyvoice/cli/frontend.py text_normalize + cosyvoice/utils/frontend_utils.py [EXT-recalled]: language by the presence of
# CJK characters; zh: newlines dropped, digits spelled out, '.' -> '。', ' - ' -> '，', bracket characters removed, a trailing '，'
# -> '。'; en: numbers spelled out; then split_paragraph (character count for zh, TOKEN count for everything else;
# token_max_n 80, token_min_n 60, merge_len 20, comma_split False) and segments that hold nothing but punctuation are dropped.
_CJK = re.compile(r"[\u4e00-\u9fff]")
_ZH_PUNC = ["。", "？", "！", "；", "：", "、", ".", "?", "!", ";"]
_EN_PUNC = [".", "?", "!", ";", ":"]
_ZH_DIGIT = "零一二三四五六七八九"
_ONLY_PUNC = re.compile(r"^[\s\.,;:!?。，、；：？！…\"'“”‘’\-—()（）\[\]【】《》<>~·]*$")


def contains_chinese(text: str) -> bool:
    return _CJK.search(text) is not None


def remove_bracket(text: str) -> str:
    for ch in "（）【】":
        text = text.replace(ch, "")
    return text.replace("`", "").replace("——", " ")


def replace_blank(text: str) -> str:
    """zh: a blank survives only between two ASCII (non-blank) characters."""
    out = []
    for i, c in enumerate(text):
        if c == " ":
            if 0 < i < len(text) - 1 and text[i + 1].isascii() and text[i + 1] != " " and text[i - 1].isascii() and text[i - 1] != " ":
                out.append(c)
        else:
            out.append(c)
    return "".join(out)


def replace_corner_mark(text: str) -> str:
    return text.replace("²", "平方").replace("³", "立方")


def _zh_number(n: int) -> str:
    """0 .. 99 999 999 in Chinese numerals (cn2an-style 'low' reading: 一十 -> 十 at the front)."""
    if n == 0:
        return _ZH_DIGIT[0]
    def below_10000(v: int) -> str:
        out, units, zero = "", ["", "十", "百", "千"], False
        for pos in (3, 2, 1, 0):
            d = v // 10 ** pos % 10
            if d == 0:
                zero = bool(out)
            else:
                out += (_ZH_DIGIT[0] if zero else "") + _ZH_DIGIT[d] + units[pos]
                zero = False
        return out
    hi, lo = divmod(n, 10000)
    s = (below_10000(hi) + "万" if hi else "") + ((_ZH_DIGIT[0] if hi and lo < 1000 and lo else "") + below_10000(lo) if lo else "")
    return s[1:] if s.startswith("一十") else s


_ONES = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen", "fourteen",
         "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]


def _en_number(n: int) -> str:
    """inflect's number_to_words for 0 .. 999 999 999 ('one hundred and five', 'twenty-one', 'one thousand, two hundred')."""
    if n < 20:
        return _ONES[n]
    if n < 100:
        return _TENS[n // 10] + ("-" + _ONES[n % 10] if n % 10 else "")
    if n < 1000:
        return _ONES[n // 100] + " hundred" + (" and " + _en_number(n % 100) if n % 100 else "")
    for unit, name in ((10 ** 6, "million"), (10 ** 3, "thousand")):
        if n >= unit:
            hi, lo = divmod(n, unit)
            if not lo:
                return _en_number(hi) + " " + name
            return _en_number(hi) + " " + name + (" and " if lo < 100 else ", ") + _en_number(lo)
    return str(n)


def spell_out_number(text: str, lang: str, speller=None) -> str:
    """Runs of digits -> words (upstream: inflect for en, cn2an inside its zh normaliser); ``speller(int) -> str`` overrides.
    Digit runs too long for the built-in spellers are read digit by digit."""
    def sub(m):
        digits = m.group(0)
        if speller is not None:
            return speller(int(digits))
        if len(digits) > 8:
            return "".join(_ZH_DIGIT[int(c)] for c in digits) if lang == "zh" else " ".join(_ONES[int(c)] for c in digits)
        return _zh_number(int(digits)) if lang == "zh" else _en_number(int(digits))
    return re.sub(r"\d+", sub, text)


def split_paragraph(text: str, tokenize, lang: str = "zh", token_max_n: int = 80, token_min_n: int = 60, merge_len: int = 20,
                    comma_split: bool = False) -> List[str]:
    """upstream frontend_utils.split_paragraph: cut after sentence punctuation (a closing quote stays with its sentence), then pack
    the sentences into segments of at most ``token_max_n`` units -- CHARACTERS for zh, TOKENS otherwise -- starting a new segment
    when adding the next sentence would pass token_max_n and the segment already holds more than token_min_n; a last segment shorter
    than ``merge_len`` joins its predecessor."""
    calc = (lambda t: len(t)) if lang == "zh" else (lambda t: len(tokenize(t)))
    punc = list(_ZH_PUNC if lang == "zh" else _EN_PUNC)
    if comma_split:
        punc += ["，", ","]
    if text and text[-1] not in punc:
        text += "。" if lang == "zh" else "."
    utts, st = [], 0
    for i, c in enumerate(text):
        if c in punc:
            if len(text[st:i]) > 0:
                utts.append(text[st:i] + c)
            if i + 1 < len(text) and text[i + 1] in ('"', "”"):
                tmp = utts.pop(-1) if utts else ""
                utts.append(tmp + text[i + 1])
                st = i + 2
            else:
                st = i + 1
    final, cur = [], ""
    for u in utts:
        if calc(cur + u) > token_max_n and calc(cur) > token_min_n:
            final.append(cur)
            cur = ""
        cur = cur + u
    if len(cur) > 0:
        if calc(cur) < merge_len and final:
            final[-1] = final[-1] + cur
        else:
            final.append(cur)
    return final


def text_normalize(text: str, tokenizer, split: bool = True, token_max_n: int = 80, token_min_n: int = 60,
                   merge_len: int = 20, speller=None) -> List[str]:
    """cosyvoice frontend.text_normalize(split=True) [EXT-recalled], without the optional WeTextProcessing / ttsfrd normalisers
    (not available offline; upstream itself falls back to this path without them): language-dependent clean-up and number
    spelling, then ``split_paragraph`` on characters (zh) or tokens (other languages); segments of punctuation only are dropped."""
    text = text.strip()
    if not text:
        return [text]
    if contains_chinese(text):
        lang = "zh"
        text = text.replace("\n", "")
        text = replace_blank(text)
        text = replace_corner_mark(text)
        text = spell_out_number(text, "zh", speller)
        text = text.replace(".", "。").replace(" - ", "，")
        text = remove_bracket(text)
        text = re.sub(r"[，,、]+$", "。", text)
    else:
        lang = "en"
        text = spell_out_number(text, "en", speller)
    if not split:
        return [text]
    segs = split_paragraph(text, tokenizer.encode, lang, token_max_n, token_min_n, merge_len, comma_split=False)
    segs = [s for s in segs if not _ONLY_PUNC.match(s)]
    return segs or [text]


class _Lazy:
    """A frontend network built on first use (seeded synthetic weights at the published shapes cost seconds to draw and pack; a run
    that injects its features never pays for them)."""

    def __init__(self, factory, synthetic: bool, build_now: bool = False):
        self._factory, self._obj, self.synthetic = factory, None, synthetic
        if build_now:
            self.get()

    def get(self):
        if self._obj is None:
            self._obj = self._factory()
        return self._obj

    def __call__(self, wav16k: torch.Tensor) -> torch.Tensor:
        return self.get()(wav16k)


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

    # ------------------------------------------------------------------ what CosyVoice(model_dir) builds
    @classmethod
    def from_model_dir(cls, model_dir: str, cfg: SynthConfig, device, weights_loaded: bool, allow_standins: Optional[bool] = None,
                       seed: int = 0) -> "Frontend":
        """The frontend ``CosyVoice(model_dir)`` loads beside llm.pt / flow.pt / hift.pt (/root/reference/tts_with_rag.py:159 [EXT]:
        tokenizer, speech_tokenizer_v1.onnx, campplus.onnx):
          text tokenizer     ``model_dir/*.tiktoken`` (or ``model_dir/tokenizer/*.tiktoken``) -> astts.bpe.TiktokenBPE
          speech tokenizer   ``model_dir/speech_tokenizer_v1.{pt,onnx}`` -> astts.frontend_nets.SpeechTokenizerV1
          speaker network    ``model_dir/campplus.{pt,onnx}``            -> astts.frontend_nets.CamPlusSpeakerNet
        A part whose file is absent falls back -- the networks to seeded SYNTHETIC weights at the published shapes (built on first
        use), the tokenizer to the byte-level stand-in.  With ``weights_loaded`` (real llm / flow / hift weights) any such fallback
        RAISES unless ``allow_standins`` (default: ASTTS_ALLOW_STANDIN_FRONTEND=1): trained networks fed untrained features are
        garbage in, garbage out, with status OK."""
        import glob
        import os

        from . import frontend_nets as fn
        from . import frontend_weights as fw

        tshape, cshape = fn.shapes_for(cfg)
        missing = []
        tik = sorted(glob.glob(os.path.join(model_dir, "*.tiktoken")) + glob.glob(os.path.join(model_dir, "tokenizer", "*.tiktoken")))
        if tik:
            from .bpe import TiktokenBPE
            tokenizer = TiktokenBPE.from_file(tik[0])
            if tokenizer.n_vocab > cfg.text_vocab:
                raise ValueError(f"{tik[0]}: {tokenizer.n_vocab} tokens, the model's text embedding has {cfg.text_vocab} rows")
        else:
            tokenizer = ByteTokenizer(cfg.text_vocab)
            missing.append("*.tiktoken (text tokenizer vocabulary)")
        parts = {}
        for stem, shape, maker, klass in (("speech_tokenizer_v1", tshape, fw.make_speech_tokenizer_weights, fn.SpeechTokenizerV1),
                                          ("campplus", cshape, fw.make_campplus_weights, fn.CamPlusSpeakerNet)):
            have = any(os.path.exists(os.path.join(model_dir, stem + ext)) for ext in (".pt", ".onnx"))
            if have:
                def load(stem=stem, shape=shape, maker=maker, klass=klass):
                    want = fw.manifest_for(maker, shape)
                    return klass(fw.load_frontend_weights(model_dir, stem, want), shape, device)
                parts[stem] = _Lazy(load, synthetic=False, build_now=True)
            else:
                missing.append(f"{stem}.pt / {stem}.onnx")
                parts[stem] = _Lazy(lambda shape=shape, maker=maker, klass=klass: klass(maker(shape, seed), shape, device, synthetic=True),
                                    synthetic=True)
        if weights_loaded and missing:
            if allow_standins is None:
                allow_standins = os.environ.get("ASTTS_ALLOW_STANDIN_FRONTEND") == "1"
            if not allow_standins:
                raise FileNotFoundError(
                    f"CosyVoice: {model_dir!r} holds llm.pt / flow.pt / hift.pt but not {', '.join(missing)}.  The loaded networks would be fed "
                    f"byte-level text ids / tokens and speaker vectors of untrained stand-ins: garbage audio with status OK.  Put the files "
                    f"there (the reference's model directory has them: tts_with_rag.py:159), inject precomputed features "
                    f"(Frontend(features=...)), or pass allow_standin_frontend=True / ASTTS_ALLOW_STANDIN_FRONTEND=1 to run anyway.")
        return cls(cfg, tokenizer=tokenizer, speech_tokenizer=parts["speech_tokenizer_v1"], speaker_embedder=parts["campplus"], device=device)

    def describe(self) -> Dict[str, str]:
        """Which implementation each learned part is: 'loaded' (weights / vocabulary from model_dir), 'synthetic-weights' (the real
        network on seeded random weights), 'stand-in' (a labelled placeholder), 'custom' (injected by the caller)."""
        def kind(p):
            if getattr(p, "standin", False):
                return "stand-in"
            if hasattr(p, "synthetic"):
                return "synthetic-weights" if p.synthetic else "loaded"
            return "loaded" if type(p).__name__ == "TiktokenBPE" else "custom"
        return {"text_tokenizer": kind(self.tokenizer), "speech_tokenizer": kind(self.speech_tokenizer),
                "speaker_embedder": kind(self.speaker_embedder)}

    def text_ids(self, text: str) -> torch.Tensor:
        ids = self.tokenizer.encode(text) or [0]
        return torch.tensor([ids], dtype=torch.int64)

    def prompts(self, wavs: List[torch.Tensor]) -> List[PromptFeatures]:
        """``prompt`` for many 16 kHz prompts at once: prompts of ONE length go through the GPU frontend as a batch (one pass of the
        speech tokenizer, one of the speaker network -- its ~400 small launches are the same for 1 and for 16 prompts -- one resample +
        log-mel) and come back with one host synchronisation per group instead of three per prompt.  Neither network pads or masks
        inside an equal-length batch, so a row differs from ``prompt(wav)`` only by the summation order of the GEMM tiles its row
        count selects (speaker vectors to ~1e-6, a speech token only at a near-tie of the codebook search).  Stand-in / injected
        parts fall back to the per-prompt path."""
        st, se = self.speech_tokenizer, self.speaker_embedder
        st = st.get() if isinstance(st, _Lazy) else st
        se = se.get() if isinstance(se, _Lazy) else se
        if self.device is None or not hasattr(st, "tokens_device") or not hasattr(se, "embed_device"):
            return [self.prompt(w) for w in wavs]
        cfg = self.cfg
        out: List[Optional[PromptFeatures]] = [None] * len(wavs)
        groups: Dict[int, List[int]] = {}
        for i, w in enumerate(wavs):
            groups.setdefault(int(w.shape[-1]), []).append(i)

        def features(batch):        # [B, n] on the GPU -> (tokens [B, T'], speaker vectors [B, spk], prompt mel [B, Tm, mel]) on the GPU
            tok = st.tokens_device(batch)
            emb = se.embed_device(batch)
            wav_sr = audio.resample(batch, 16000, cfg.sample_rate)
            mel = audio.mel_spectrogram(wav_sr, sr=cfg.sample_rate, n_fft=1024, hop=cfg.hop, win=1024, n_mels=cfg.mel, fmin=0.0, fmax=8000.0)
            return tok, emb, mel

        for n, idxs in groups.items():
            batch = torch.cat([wavs[i].reshape(1, -1) for i in idxs], 0).to(self.device, non_blocking=True)
            tok, emb, mel = features(batch)
            n_tok = min(tok.shape[1], int(mel.shape[1] * cfg.token_rate * cfg.hop / cfg.sample_rate))
            n_mel = cfg.mel_frames_for_tokens(n_tok)
            tok_h, emb_h, mel_h = tok[:, :n_tok].cpu(), emb.cpu(), mel[:, :n_mel].cpu()
            for j, i in enumerate(idxs):
                out[i] = PromptFeatures(tok_h[j:j + 1].contiguous(), emb_h[j:j + 1].contiguous(), mel_h[j:j + 1].contiguous())
        return out

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
