"""Real-checkpoint readiness without a checkpoint (the reference loads CosyVoice-300M from a directory: /root/reference/tts_with_rag.py:159):
the expected-key / shape manifest derived from SynthConfig, the loader's one-error report, weight_norm in both of torch's forms, the
extra buffers upstream checkpoints carry, and the plain-JSON model config ``model_dir/astts.json``.  CPU only: state dicts are
synthetic (tiny shapes) but carry upstream's exact key set."""
import json
import os
import warnings

import pytest
import torch

from astts.synth import weights as W
from astts.synth.config import SynthConfig


def _as_checkpoint(sd, form):
    """Re-express every conv / transposed-conv weight of a state dict the way torch.nn.utils.weight_norm saves it (form 0: weight_g /
    weight_v; form 1: parametrizations.weight.original0 / original1), with g != |v| so that folding is not the identity."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() == 3:
            base = k[: -len(".weight")]
            norm = v.flatten(1).norm(dim=1).view(-1, 1, 1)
            vv = v * 1.7                                              # any direction-preserving rescale: g restores the magnitude
            g = norm.clone()
            names = (".weight_g", ".weight_v") if form == 0 else (".parametrizations.weight.original0", ".parametrizations.weight.original1")
            out[base + names[0]], out[base + names[1]] = g, vv
        else:
            out[k] = v
    return out


def _write_model_dir(path, cfg, state, forms=(0, 1), extras=True, cfg_json=True):
    os.makedirs(path, exist_ok=True)
    llm, flow, hift = dict(state["llm"]), dict(state["flow"]), _as_checkpoint(state["hift"], forms[0])
    flow = {k: v for k, v in _as_checkpoint(flow, forms[1]).items()}
    if extras:                                                        # registered buffers / counters real checkpoints carry
        hift["stft_window"] = torch.hann_window(16)
        llm["text_encoder.embed.pos_enc.pe"] = torch.zeros(1, 8, cfg.lm_dim)
        flow["encoder.embed.pos_enc.pe"] = torch.zeros(1, 8, cfg.flow_dim)
        hift["m_source.l_sin_gen.harmonics"] = torch.arange(9.0)
    torch.save(llm, os.path.join(path, "llm.pt"))
    torch.save(flow, os.path.join(path, "flow.pt"))
    torch.save(hift, os.path.join(path, "hift.pt"))
    if cfg_json:
        with open(os.path.join(path, "astts.json"), "w") as f:
            f.write(cfg.to_json())


def test_manifest_matches_the_synthetic_weights_at_both_sizes():
    for cfg in (SynthConfig.tiny(), SynthConfig.tiny().with_(up_rates=(8, 4), sample_rate=24000)):
        want, sd = W.expected_shapes(cfg), W.make_all(cfg, 0)
        for part in ("llm", "flow", "hift"):
            assert set(want[part]) == set(sd[part])
            assert all(tuple(sd[part][k].shape) == tuple(want[part][k]) for k in want[part])
    full = W.expected_shapes(SynthConfig())                           # CosyVoice-300M: shapes only, nothing allocated
    n = {p: sum(int(torch.Size(s).numel()) for s in full[p].values()) for p in full}
    assert 300e6 < n["llm"] < 320e6 and 100e6 < n["flow"] < 110e6 and 19e6 < n["hift"] < 22e6
    assert full["llm"]["llm_decoder.weight"] == (4097, 1024) and full["hift"]["ups.0.weight"] == (512, 256, 16)


def test_loader_folds_weight_norm_ignores_known_buffers_and_reads_the_json_config(tmp_path):
    cfg = SynthConfig.tiny().with_(sample_rate=24000, max_positions=640)
    state = W.make_all(cfg, 3)
    d = str(tmp_path / "CosyVoice-300M")
    _write_model_dir(d, cfg, state)
    assert SynthConfig.from_json(os.path.join(d, "astts.json")) == cfg
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                # the known extra buffers raise no warning
        got = W.load_state_dicts(d, cfg)
    for part in ("llm", "flow", "hift"):
        for k, v in state[part].items():
            assert torch.allclose(got[part][k], v, rtol=1e-5, atol=1e-6), (part, k)     # folded back to the plain weight
    # an unknown extra tensor is reported (warning), and is an error under strict
    hift = torch.load(os.path.join(d, "hift.pt"), weights_only=True)
    hift["something.new"] = torch.zeros(3)
    torch.save(hift, os.path.join(d, "hift.pt"))
    with pytest.warns(RuntimeWarning, match="something.new"):
        W.load_state_dicts(d, cfg)
    with pytest.raises(ValueError, match="something.new"):
        W.load_state_dicts(d, cfg, strict=True)


def test_loader_reports_missing_and_misshaped_keys_in_one_error(tmp_path):
    cfg = SynthConfig.tiny()
    state = W.make_all(cfg, 4)
    del state["llm"]["llm.encoders.1.self_attn.linear_pos.weight"]
    del state["flow"]["decoder.estimator.final_proj.bias"]
    state["hift"]["conv_post.bias"] = torch.zeros(17)
    d = str(tmp_path / "broken")
    _write_model_dir(d, cfg, state, extras=False)
    with pytest.raises(ValueError) as e:
        W.load_state_dicts(d, cfg)
    msg = str(e.value)
    assert "llm.encoders.1.self_attn.linear_pos.weight" in msg and "decoder.estimator.final_proj.bias" in msg
    assert "conv_post.bias is (17,), expected (18,)" in msg
    # the wrong CONFIG for a good checkpoint reads the same way (widths of the 300M model against tiny tensors)
    good = str(tmp_path / "good")
    _write_model_dir(good, cfg, W.make_all(cfg, 4), cfg_json=False)
    with pytest.raises(ValueError, match="mis-shaped"):
        W.load_state_dicts(good, SynthConfig())


def test_json_config_rejects_unknown_fields_and_round_trips():
    cfg = SynthConfig().with_(sample_rate=24000, max_positions=8192, up_rates=(8, 8))
    assert SynthConfig.from_json(cfg.to_json()) == cfg
    assert SynthConfig.from_json('{"sample_rate": 24000}').sample_rate == 24000
    with pytest.raises(ValueError, match="sample_rat"):
        SynthConfig.from_json(json.dumps({"sample_rat": 24000}))
    with pytest.raises(ValueError, match="eos_policy"):
        SynthConfig.from_json(json.dumps({"eos_policy": "maybe"}))


def test_cosyvoice_constructor_reads_the_config_and_checks_the_checkpoint_before_it_needs_a_gpu(tmp_path):
    from astts.compat.cosyvoice import CosyVoice

    cfg = SynthConfig.tiny().with_(sample_rate=24000)
    d = str(tmp_path / "m")
    _write_model_dir(d, cfg, W.make_all(cfg, 5))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="ROCm GPU"):           # config read, checkpoint loaded and checked: only the engine is missing
            CosyVoice(d)
    os.remove(os.path.join(d, "astts.json"))                          # without it the 300M defaults apply: the tiny checkpoint does not fit them
    with pytest.raises(ValueError, match="does not match the model config"):
        CosyVoice(d)
