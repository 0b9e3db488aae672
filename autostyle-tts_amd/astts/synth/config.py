"""Architecture hyper-parameters of the synthesis engine behind the ``CosyVoice`` call surface.

Defaults restate the published CosyVoice-300M configuration (the model every reference script loads:
/root/reference/tts_with_rag.py:159, tts_with_style_and_timbre.py:74).  The reference's own copy of
that code is a private fork that is not vendored, so these numbers are [EXT]-recalled (SURVEY.md 8a
rows a13-a15) and every one of them is a field, not a constant.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Tuple


@dataclass(frozen=True)
class SynthConfig:
    sample_rate: int = 22050          # what the scripts save (tts_with_rag.py:197); 24000 for the README figure
    # ---- acoustic transformer (TransformerLM)
    text_vocab: int = 51866
    text_dim: int = 512
    speech_vocab: int = 4096          # + 1 EOS logit
    spk_dim: int = 192
    lm_dim: int = 1024
    lm_heads: int = 16
    lm_ffn: int = 4096
    lm_text_layers: int = 6
    lm_layers: int = 14
    top_k: int = 25
    top_p: float = 0.8
    ras_win: int = 10
    ras_tau: float = 0.1
    # While a row may not stop yet (upstream: the first min_len = 2 x text-length steps), "reject" = upstream's sampling_ids: sample
    # again until the token is not EOS (EOS keeps its probability, its nucleus slot and its share of top_p); "mask" = remove the EOS
    # logit before the softmax (this build's rounds 1-3).  Different distributions inside that window: DESIGN.md section 2.
    eos_policy: str = "reject"
    # ---- flow-matching decoder (MaskedDiffWithXvec + ConditionalCFM)
    flow_dim: int = 512
    flow_heads: int = 8
    flow_ffn: int = 2048
    flow_layers: int = 6
    mel: int = 80
    token_rate: int = 50              # speech tokens per second
    hop: int = 256                    # samples per mel frame
    est_channels: Tuple[int, ...] = (256, 256)
    est_mid_blocks: int = 12
    est_tfm_per_block: int = 4
    est_heads: int = 8                # x 64 head dim
    est_groups: int = 8
    cfm_steps: int = 10
    cfg_rate: float = 0.7
    # ---- HiFT vocoder
    hift_base: int = 512
    up_rates: Tuple[int, ...] = (8, 8)
    res_kernels: Tuple[int, ...] = (3, 7, 11)
    res_dils: Tuple[int, ...] = (1, 3, 5)
    src_res_kernels: Tuple[int, ...] = (7, 11)
    nb_harmonics: int = 8
    f0_channels: int = 512
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    nsf_voiced_threshold: float = 10.0
    lrelu_slope: float = 0.1
    audio_limit: float = 0.99
    ln_eps: float = 1e-5
    # Size of the per-layer relative-position tables (rows for rel = -max_positions .. max_positions): an LM context (prefix +
    # generated tokens) may not exceed it.  Upstream's espnet encoding extends on demand; here the tables are formed at load, sized
    # for a 30 s style prompt (1 500 tokens) + an 80-token segment's 20x window (1 600) + text, and the drop-in surface WARNS when a
    # request is capped by them (compat/cosyvoice.py); pass a larger value for longer contexts (117 MB of fp16 tables per 2 048).
    max_positions: int = 4096

    def __post_init__(self):
        if self.eos_policy not in ("mask", "reject"):
            raise ValueError(f"SynthConfig.eos_policy {self.eos_policy!r}: expected 'mask' or 'reject'")

    @property
    def est_time_dim(self) -> int:
        return self.est_channels[0] * 4

    @property
    def est_in(self) -> int:
        return 4 * self.mel            # x | mu | spk | cond

    @property
    def upsample_total(self) -> int:
        n = 4                          # iSTFT hop
        for r in self.up_rates:
            n *= r
        return n

    def mel_frames_for_tokens(self, n_tokens: int) -> int:
        return int(n_tokens / self.token_rate * self.sample_rate / self.hop)

    @staticmethod
    def tiny() -> "SynthConfig":
        """Same topology, toy widths: for parity tests that run in seconds on CPU."""
        return SynthConfig(text_vocab=300, text_dim=64, speech_vocab=256, spk_dim=32, lm_dim=128, lm_heads=2,
                           lm_ffn=256, lm_text_layers=2, lm_layers=2, flow_dim=128, flow_heads=2, flow_ffn=256,
                           flow_layers=2, est_channels=(64, 64), est_mid_blocks=2, est_tfm_per_block=1, est_heads=2,
                           hift_base=64, f0_channels=64, max_positions=512)

    def with_(self, **kw) -> "SynthConfig":
        return replace(self, **kw)

    # ---- model_dir/astts.json: the plain-JSON model config of this build (upstream keeps its hyper-parameters in
    # model_dir/cosyvoice.yaml, a hyperpyyaml file that instantiates Python classes -- not readable here and not a data format)
    def to_json(self) -> str:
        import dataclasses
        import json
        return json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in dataclasses.asdict(self).items()}, indent=1)

    @staticmethod
    def from_json(text_or_path: str) -> "SynthConfig":
        """A JSON object whose keys are SynthConfig fields (any subset: the rest keep the CosyVoice-300M defaults), given as text or
        as a file path.  Unknown keys are an error (a typo must not silently fall back to a default)."""
        import dataclasses
        import json
        import os
        text = text_or_path
        if not text_or_path.lstrip().startswith("{") and os.path.exists(text_or_path):
            with open(text_or_path, "r", encoding="utf-8") as f:
                text = f.read()
        data = json.loads(text)
        fields = {f.name: f for f in dataclasses.fields(SynthConfig)}
        unknown = sorted(set(data) - set(fields))
        if unknown:
            raise ValueError(f"astts.json: unknown SynthConfig field(s) {unknown}; known: {sorted(fields)}")
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in data.items()}
        return SynthConfig(**kw)
