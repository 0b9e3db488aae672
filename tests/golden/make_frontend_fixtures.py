"""Generates tests/golden/frontend_blocks.npz: inputs and expected outputs of the speech tokenizer's encoder, produced by an
independent third-party implementation of the same published architecture -- transformers' ``WhisperEncoder`` /
``WhisperEncoderLayer`` (the reference runs CosyVoice's speech_tokenizer_v1.onnx, a Whisper-style encoder + codebook, inside
CosyVoice(model_dir): /root/reference/tts_with_rag.py:159,195) -- on the seeded synthetic weights of
astts.frontend_weights.make_speech_tokenizer_weights.  Run in the BUILD container only
(python tests/golden/make_frontend_fixtures.py); the .npz is data: mel frames in, hidden states out.  Nothing of transformers
travels; the weights regenerate from their seed.

Pins: conv stem + sinusoidal positions (``encoder.embed_positions`` as transformers initialises it), one encoder layer with and
without a key-padding mask, the full stack in front of transformers' final LayerNorm (the tokenizer has none)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "autostyle-tts_amd")]

from transformers import WhisperConfig  # noqa: E402
from transformers.models.whisper.modeling_whisper import WhisperEncoder  # noqa: E402

from astts.frontend_weights import SpeechTokenizerShape, make_speech_tokenizer_weights  # noqa: E402

SEED = 31


def hf_state(sd, cfg):
    out = {"conv1.weight": sd["encoder.conv1.weight"], "conv1.bias": sd["encoder.conv1.bias"],
           "conv2.weight": sd["encoder.conv2.weight"], "conv2.bias": sd["encoder.conv2.bias"]}
    names = {"attn_ln": "self_attn_layer_norm", "attn.query": "self_attn.q_proj", "attn.key": "self_attn.k_proj",
             "attn.value": "self_attn.v_proj", "attn.out": "self_attn.out_proj", "mlp_ln": "final_layer_norm", "mlp.0": "fc1",
             "mlp.2": "fc2"}
    for i in range(cfg.layers):
        for a, b in names.items():
            for leaf in ("weight", "bias"):
                k = f"encoder.blocks.{i}.{a}.{leaf}"
                if k in sd:
                    out[f"layers.{i}.{b}.{leaf}"] = sd[k]
    return out


def main():
    cfg = SpeechTokenizerShape.tiny()
    t_mel = 2 * 90
    sd = make_speech_tokenizer_weights(cfg, SEED)
    hf = WhisperConfig(num_mel_bins=cfg.n_mels, d_model=cfg.d, encoder_layers=cfg.layers, encoder_attention_heads=cfg.heads,
                       encoder_ffn_dim=4 * cfg.d, max_source_positions=t_mel // 2, activation_function="gelu", dropout=0.0,
                       attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0)
    hf._attn_implementation = "eager"
    enc = WhisperEncoder(hf).eval().float()
    positions = enc.embed_positions.weight.detach().clone()          # transformers' own sinusoids
    missing = enc.load_state_dict(hf_state(sd, cfg), strict=False)
    assert set(missing.missing_keys) <= {"embed_positions.weight", "layer_norm.weight", "layer_norm.bias"} and not missing.unexpected_keys, missing
    g = torch.Generator().manual_seed(SEED + 1)
    mel = torch.randn(2, cfg.n_mels, t_mel, generator=g)
    out = {"seed": np.int64(SEED), "mel": mel.numpy(), "positions": positions.numpy()}
    states = []
    hooks = [layer.register_forward_hook(lambda m, a, o: states.append((o[0] if isinstance(o, tuple) else o).detach().clone()))
             for layer in enc.layers]
    pre = []
    hooks.append(enc.layers[0].register_forward_pre_hook(lambda m, a: pre.append(a[0].detach().clone())))
    with torch.no_grad():
        enc(mel)
        out["stem"] = pre[0].numpy()                                  # conv1 -> gelu -> conv2 -> gelu -> + positions
        for i, s in enumerate(states):
            out[f"layer{i}"] = s.numpy()
        # one layer with a key-padding mask (additive -inf on the padded keys), rows of different lengths
        lens = torch.tensor([t_mel // 2, 37])
        valid = torch.arange(t_mel // 2)[None, :] < lens[:, None]
        add = torch.zeros(2, 1, t_mel // 2, t_mel // 2).masked_fill(~valid[:, None, None, :], float("-inf"))
        o = enc.layers[0](pre[0], add)
        out["layer0_masked"] = (o[0] if isinstance(o, tuple) else o).numpy()
        out["masked_lens"] = lens.numpy()
    for h in hooks:
        h.remove()
    path = os.path.join(ROOT, "tests", "golden", "frontend_blocks.npz")
    np.savez_compressed(path, **out)
    print("->", path, os.path.getsize(path) // 1024, "KB", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
