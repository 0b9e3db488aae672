"""oracle/synth.py pinned block by block to independent third-party implementations of the published blocks it restates
(tests/golden/make_synth_block_fixtures.py: transformers' FastSpeech2-Conformer / Wav2Vec2-Conformer relative-position
attention, dac's Snake, SpeechT5's HiFi-GAN residual block and generator trunk, Whisper's feature extractor).  The reference's
own synthesis code is an un-vendored private fork (/root/reference/tts_with_rag.py:1-2,18-19,159,195), so this is the evidence
available that the oracle restates the published algorithms; what stays [EXT]-recalled is listed in DESIGN.md section 2.
fp32 vs fp32: 2e-5 of the tensor's scale."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import synth as osyn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLD, "synth_blocks.npz"))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def _t(x):
    return torch.from_numpy(np.asarray(x))


def _attn_sd(fx, pre):
    names = ["linear_q.weight", "linear_q.bias", "linear_k.weight", "linear_k.bias", "linear_v.weight", "linear_v.bias",
             "linear_out.weight", "linear_out.bias", "linear_pos.weight", "pos_bias_u", "pos_bias_v"]
    return {"a." + n: _t(fx[pre + n]) for n in names}


@pytest.mark.parametrize("pre", ["fs2.", "w2v."])
def test_relpos_attention_matches_two_third_party_implementations(fx, pre):
    """RelPositionMultiHeadedAttention: q/k/v/pos projections, pos_bias_u / pos_bias_v, the relative shift, masking, softmax,
    output projection -- against FastSpeech2ConformerAttention (with a key-length mask) and Wav2Vec2ConformerSelfAttention."""
    sd = _attn_sd(fx, pre)
    x = _t(fx[pre + "x"])
    b, t, d = x.shape
    heads = int(fx[pre + "heads"])
    lens = _t(fx[pre + "lens"]) if pre + "lens" in fx.files else torch.full((b,), t)
    # the position table: the third party's rows run from relative position +(t-1) down to -(t-1); the oracle indexes rel + center
    pe = osyn.rel_pos_table(d, t + 3)
    theirs = _t(fx[pre + "pos_emb"])[0]
    mine = pe[(t + 3) - (t - 1):(t + 3) + t].flip(0)
    assert mine.shape == theirs.shape and _rel(mine, theirs) < 1e-6
    y, _ = osyn.relpos_attention(sd, "a", x, heads, pe, t + 3, lens, causal=False)
    ref = _t(fx[pre + "y"])
    for i in range(b):                       # rows beyond a sequence's length are padding on both sides
        n = int(lens[i])
        assert _rel(y[i, :n], ref[i, :n]) < 2e-5, (pre, i)


def test_relpos_attention_causal_cache_equals_full_pass(fx):
    """The incremental form the LM decode uses (new positions against cached keys) equals the full causal pass."""
    sd = _attn_sd(fx, "fs2.")
    x = _t(fx["fs2.x"])
    b, t, d = x.shape
    heads = int(fx["fs2.heads"])
    pe = osyn.rel_pos_table(d, 64)
    lens = torch.full((b,), t)
    full, _ = osyn.relpos_attention(sd, "a", x, heads, pe, 64, lens, causal=True)
    y0, cache = osyn.relpos_attention(sd, "a", x[:, :20], heads, pe, 64, torch.full((b,), 20), causal=True)
    y1, _ = osyn.relpos_attention(sd, "a", x[:, 20:], heads, pe, 64, lens, causal=True, cache=cache)
    assert _rel(torch.cat([y0, y1], 1), full) < 1e-5


def test_snake_matches_dac(fx):
    x = _t(fx["snake.x"]).transpose(1, 2)                 # the oracle is channels-last
    y = osyn._snake(x, _t(fx["snake.alpha"]))
    assert _rel(y.transpose(1, 2), fx["snake.y"]) < 1e-6


def _resblock_sd(fx, pre, p, channels):
    sd = {}
    for j in range(3):
        for c in ("convs1", "convs2"):
            sd[f"{p}.{c}.{j}.weight"] = _t(fx[f"{pre}{c}.{j}.weight"])
            sd[f"{p}.{c}.{j}.bias"] = _t(fx[f"{pre}{c}.{j}.bias"])
        sd[f"{p}.activations1.{j}.alpha"] = torch.ones(channels)
        sd[f"{p}.activations2.{j}.alpha"] = torch.ones(channels)
    return sd


@pytest.mark.parametrize("k", [3, 7, 11])
def test_resblock_structure_matches_speecht5_hifigan(fx, k, monkeypatch):
    """HiFi-GAN ResBlock1: per dilation d, act -> Conv1d(k, dilation d, padding d (k - 1) / 2) -> act -> Conv1d(k, padding
    (k - 1) / 2) -> + residual.  The third-party block uses leaky-relu where HiFT uses Snake (pinned separately), so the oracle's
    activation is swapped for the comparison: everything else -- paddings, dilations, residual order -- is the oracle's code."""
    monkeypatch.setattr(osyn, "_snake", lambda x, alpha: F.leaky_relu(x, 0.1))
    sd = _resblock_sd(fx, f"rb{k}.", "rb", 12)
    y = osyn._resblock(sd, "rb", _t(fx[f"rb{k}.x"]).transpose(1, 2), k, (1, 3, 5))
    assert _rel(y.transpose(1, 2), fx[f"rb{k}.y"]) < 2e-5


def test_hift_trunk_matches_speecht5_hifigan_generator(fx, monkeypatch):
    """conv_pre -> [leaky-relu(0.1) -> ConvTranspose1d(16, stride 8, padding 4) -> mean of the three parallel resblocks] x 2 ->
    leaky-relu(0.01), with the source branch silenced (zero source_downs / source_resblocks) and leaky-relu for Snake.  HiFT
    reflect-pads one sample on the left before its last stage (the iSTFT head needs L / 4 + 1 frames), so the oracle's frame t + 1
    is the third party's frame t; compared away from the edges (receptive field of the last stage: 60 frames)."""
    from astts.synth.config import SynthConfig

    monkeypatch.setattr(osyn, "_snake", lambda x, alpha: F.leaky_relu(x, 0.1))
    cfg = SynthConfig.tiny()
    assert cfg.up_rates == (8, 8) and cfg.res_kernels == (3, 7, 11) and cfg.res_dils == (1, 3, 5) and cfg.lrelu_slope == 0.1
    sd = {"conv_pre.weight": _t(fx["gan.conv_pre.weight"]), "conv_pre.bias": _t(fx["gan.conv_pre.bias"])}
    ch = [16, 8]
    for i in range(2):
        sd[f"ups.{i}.weight"] = _t(fx[f"gan.upsampler.{i}.weight"])
        sd[f"ups.{i}.bias"] = _t(fx[f"gan.upsampler.{i}.bias"])
        for kk in range(3):
            sd.update(_resblock_sd(fx, f"gan.resblocks.{i * 3 + kk}.", f"resblocks.{i * 3 + kk}", ch[i]))
        kd = 16 if i == 0 else 1                                        # HiFT's source_downs: stride = remaining up-sampling
        sd[f"source_downs.{i}.weight"] = torch.zeros(ch[i], 18, kd)
        sd[f"source_downs.{i}.bias"] = torch.zeros(ch[i])
        ks = cfg.src_res_kernels[i]
        for j in range(3):
            for c in ("convs1", "convs2"):
                sd[f"source_resblocks.{i}.{c}.{j}.weight"] = torch.zeros(ch[i], ch[i], ks)
                sd[f"source_resblocks.{i}.{c}.{j}.bias"] = torch.zeros(ch[i])
            sd[f"source_resblocks.{i}.activations1.{j}.alpha"] = torch.ones(ch[i])
            sd[f"source_resblocks.{i}.activations2.{j}.alpha"] = torch.ones(ch[i])
    mel = _t(fx["gan.mel"])
    frames = mel.shape[1] * 64
    s_stft = torch.zeros(mel.shape[0], frames + 1, 18)
    x = osyn.hift_trunk(sd, cfg, mel, s_stft)                           # [B, frames + 1, C]
    ref = _t(fx["gan.pre_post"]).transpose(1, 2)                        # [B, frames, C]
    assert x.shape[1] == frames + 1
    lo, hi = 64, frames - 64
    assert _rel(x[:, lo + 1:hi + 1], ref[:, lo:hi]) < 2e-5


def test_whisper_log_mel_and_filterbank_match_the_feature_extractor(fx):
    from astts import audio

    fb = audio.mel_filterbank(16000, 400, 128, 0.0, 8000.0)            # [128, 201]
    assert _rel(fb.T, fx["whisper.mel_filters"]) < 1e-5
    feats = audio.whisper_log_mel(_t(fx["whisper.wav"]))[0].numpy()
    ref = fx["whisper.features"]
    assert feats.shape == ref.shape == (128, 200)
    assert float(np.abs(feats - ref).max()) < 2e-4                      # log10 of fp32 power sums, (x + 4) / 4


def test_encoder_layer_matches_fastspeech2_conformer_layer(fx):
    """The pre-norm layer the text encoder, the token encoder and the LM body stack: h += attn(LN(h)); h += W2 relu(W1 LN(h)),
    against FastSpeech2ConformerEncoderLayer configured without macaron feed-forward and convolution module (its position-wise
    Conv1d of kernel size 1 is the Linear)."""
    sd = {}
    for n in ("linear_q.weight", "linear_q.bias", "linear_k.weight", "linear_k.bias", "linear_v.weight", "linear_v.bias",
              "linear_out.weight", "linear_out.bias", "linear_pos.weight", "pos_bias_u", "pos_bias_v"):
        sd["L.self_attn." + n] = _t(fx["enc.self_attn." + n])
    sd["L.norm1.weight"], sd["L.norm1.bias"] = _t(fx["enc.self_attn_layer_norm.weight"]), _t(fx["enc.self_attn_layer_norm.bias"])
    sd["L.norm2.weight"], sd["L.norm2.bias"] = _t(fx["enc.ff_layer_norm.weight"]), _t(fx["enc.ff_layer_norm.bias"])
    sd["L.feed_forward.w_1.weight"], sd["L.feed_forward.w_1.bias"] = _t(fx["enc.feed_forward.conv1.weight"])[..., 0], _t(fx["enc.feed_forward.conv1.bias"])
    sd["L.feed_forward.w_2.weight"], sd["L.feed_forward.w_2.bias"] = _t(fx["enc.feed_forward.conv2.weight"])[..., 0], _t(fx["enc.feed_forward.conv2.bias"])
    x, lens = _t(fx["enc.x"]), _t(fx["enc.lens"])
    b, t, d = x.shape
    pe = osyn.rel_pos_table(d, 40)
    y, _ = osyn.relpos_layer(sd, "L", x, int(fx["enc.heads"]), pe, 40, lens, False, ("norm1", "norm2"), F.relu, 1e-5)
    ref = _t(fx["enc.y"])
    for i in range(b):
        n = int(lens[i])
        assert _rel(y[i, :n], ref[i, :n]) < 2e-5, i


@pytest.mark.parametrize("key,top_p", [("nucleus.kept_p80", 0.8), ("nucleus.kept_p50", 0.5), ("nucleus.kept_p95", 0.95)])
def test_nucleus_set_matches_transformers_top_p_warper(fx, key, top_p):
    """oracle.synth.nucleus (the candidate set of the sampler, a13) against TopPLogitsWarper: without a top_k cap the two keep the
    same tokens; with upstream's cap (25) the nucleus is the 25 most probable of that set, in (probability desc, id asc) order.
    Rows 0-11: continuous logits (flat and peaked), exact id sets.  Rows 12-15: many exactly tied logits -- which of several tied
    tokens falls inside is an ordering convention (ours: lower id first), so those compare the kept VALUES, not the ids (and allow the
    cut to fall one token apart: fp32 accumulation order)."""
    logits = torch.from_numpy(fx["nucleus.logits"])
    kept = fx[key]
    v = logits.shape[1]
    for r in range(logits.shape[0]):
        lg = logits[r]
        e = torch.exp(lg - lg.max())
        p = e * (1.0 / e.sum())
        order, cnt, cum = osyn.nucleus(p, v, top_p)
        ids = order[:cnt]
        ref = np.nonzero(kept[r])[0]
        if r < 12:
            assert sorted(ids) == ref.tolist(), (r, cnt, len(ref))
        else:
            # thousands of tied tokens deep, the fp32 running mass depends on the order of accumulation (ours descending as
            # upstream's loop, the warper's an ascending cumsum): the cut may fall one token apart
            assert abs(cnt - len(ref)) <= 1, (r, cnt, len(ref))
            n = min(cnt, len(ref))
            mine = sorted((float(lg[i]) for i in ids), reverse=True)[:n]
            theirs = sorted((float(lg[i]) for i in ref), reverse=True)[:n]
            assert mine == theirs, r
        assert float(cum) >= top_p and float(cum) - float(p[ids[-1]]) < top_p          # the shortest prefix that reaches top_p
        o25, c25, _ = osyn.nucleus(p, 25, top_p)
        assert o25[:c25] == order[:min(cnt, 25)] and c25 == min(cnt, 25)
        # rank order: probability descending, lower id first among equals
        assert all((float(p[a]), -a) >= (float(p[b]), -b) for a, b in zip(order[:cnt - 1], order[1:cnt]))


def test_estimator_transformer_block_matches_torch_prenorm_encoder_layer(fx):
    """oracle.synth._tfm_block (the BasicTransformerBlock of the flow estimator, a14: pre-LayerNorm self-attention without q/k/v bias
    + out-projection, pre-LayerNorm GELU feed-forward, two residuals, key padding mask) against
    torch.nn.TransformerEncoderLayer(norm_first=True, activation="gelu") with the same weights.  Valid positions only (what a
    padded query row holds is masked downstream)."""
    sd = {k[len("tfm."):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("tfm.") and k not in ("tfm.x", "tfm.lens", "tfm.y", "tfm.heads")}
    sd = {"blk." + k: v for k, v in sd.items()}
    x, lens, ref = torch.from_numpy(fx["tfm.x"]), torch.from_numpy(fx["tfm.lens"]), torch.from_numpy(fx["tfm.y"])
    y = osyn._tfm_block(sd, "blk", x, lens, int(fx["tfm.heads"]))
    for b in range(x.shape[0]):
        n = int(lens[b])
        err = float((y[b, :n] - ref[b, :n]).abs().max()) / float(ref[b, :n].abs().max())
        assert err < 2e-5, (b, err)


def test_prompt_mel_matches_transformers_audio_utils(fx):
    """astts.audio.mel_filterbank / mel_spectrogram (the matcha-style log-mel of the flow's prompt: 22 050 Hz, n_fft 1024, hop 256,
    80 Slaney bins over 0-8 kHz, reflect padding, natural log with floor 1e-5) against transformers.audio_utils.  The HIP kernel
    astts_op_mel_spectrogram is held to this host definition in tests/test_ops_gpu.py."""
    from astts.audio import mel_filterbank, mel_spectrogram

    fb = mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert fb.shape == fx["pmel.filters"].shape and float(np.abs(fb - fx["pmel.filters"]).max()) < 1e-7
    m = mel_spectrogram(torch.from_numpy(fx["pmel.wav"])[None])[0].numpy()
    ref = fx["pmel.logmel"]
    assert m.shape == ref.shape
    assert float(np.abs(m - ref).max()) < 1e-4, float(np.abs(m - ref).max())


def test_cfm_time_grid_and_cfg_match_transformers_dit_sampler():
    """oracle.synth.cfm_t_grid / cfg_combine (a14: the cosine-warped Euler grid and the guidance formula of ConditionalCFM) against
    what transformers' Qwen2_5OmniToken2WavDiTModel.sample hands its ODE solver at sway_coefficient = -1 and what its ode_function
    returns for recorded (conditional, unconditional) velocities -- tests/golden/cfm_grid_cfg.npz (make_synth_block_fixtures.py --cfm).
    That sampler lists `num_steps` POINTS (n - 1 intervals); the oracle's argument is the number of Euler steps."""
    fx = np.load(os.path.join(GOLD, "cfm_grid_cfg.npz"))
    for name in ("a", "b"):
        grid = fx[f"{name}.grid"]
        mine = osyn.cfm_t_grid(len(grid) - 1).numpy()
        assert mine.shape == grid.shape and float(np.abs(mine - grid).max()) < 2e-7, (name, np.abs(mine - grid).max())
        assert mine[0] == 0.0 and abs(float(mine[-1]) - 1.0) < 1e-7 and np.all(np.diff(mine) > 0)
        g = osyn.cfg_combine(torch.from_numpy(fx[f"{name}.dc"]), torch.from_numpy(fx[f"{name}.du"]), float(fx[f"{name}.scale"])).numpy()
        ref = fx[f"{name}.guided"]
        assert g.shape == ref.shape and float(np.abs(g - ref).max()) < 5e-6 * float(np.abs(ref).max()), np.abs(g - ref).max()
    # the benchmark's solve: 10 steps at rate 0.7 (SynthConfig defaults) is case "a"
    from astts.synth.config import SynthConfig

    c = SynthConfig()
    assert c.cfm_steps == len(fx["a.grid"]) - 1 and abs(c.cfg_rate - float(fx["a.scale"])) < 1e-12


def test_reject_policy_has_the_distribution_of_upstreams_redraw_loop():
    """oracle.synth.ras_sample(eos_policy="reject") takes upstream's `while True: top_ids = ras_sampling(...); if EOS not in top_ids:
    break` [EXT cosyvoice/llm/llm.py sampling_ids] in closed form with two injected uniforms.  Here the loop itself is simulated
    (fresh randomness per pass, nucleus draw -> repetition check -> full-distribution draw) and its empirical distribution is held
    against the closed form's over random (u1, u2), together with the analytic one:
        P(t) = [N(t) (1 - R(t)) + rho p(t)] / (1 - N(eos) - rho p(eos)),  N = renormalised nucleus, rho = sum of N over repeated t."""
    g = torch.Generator().manual_seed(3)
    v, eos, top_k, top_p, win, tau = 12, 11, 5, 0.8, 4, 0.25
    logits = torch.tensor([[1.2, 0.1, 2.0, -1.0, 0.7, 1.9, -0.5, 0.3, -2.0, 0.9, 0.0, 1.6]])
    hist = torch.tensor([[7, 2, 3, 5]], dtype=torch.int32)          # tokens 2 and 5 (both in the nucleus) are "repeated"
    p = torch.softmax(logits[0], 0)
    order, cnt, cum = osyn.nucleus(p, top_k, top_p)
    assert eos in order[:cnt]
    nuc = {t: float(p[t] / cum) for t in order[:cnt]}
    rep = {t for t in range(v) if t in (2, 3, 5, 7)}
    rho = sum(w for t, w in nuc.items() if t in rep)
    z = 1.0 - nuc[eos] - rho * float(p[eos])
    analytic = np.array([((nuc.get(t, 0.0) if t not in rep else 0.0) + rho * float(p[t])) / z if t != eos else 0.0 for t in range(v)])
    assert abs(analytic.sum() - 1.0) < 1e-6
    n = 40000
    # the loop, simulated
    rng = np.random.default_rng(0)
    ids = list(nuc)
    w = np.array([nuc[t] for t in ids])
    pn = p.double().numpy()
    pn = pn / pn.sum()
    counts_loop = np.zeros(v)
    for _ in range(n):
        while True:
            t = ids[rng.choice(len(ids), p=w / w.sum())]
            if t in rep:
                t = int(rng.choice(v, p=pn))
            if t != eos:
                break
        counts_loop[t] += 1
    # the closed form over random uniforms
    counts_cf = np.zeros(v)
    us = torch.rand(n, 1, 2, generator=g)
    for i in range(n):
        counts_cf[int(osyn.ras_sample(logits, hist, us[i], top_k, top_p, win, tau, eos, True, "reject")[0])] += 1
    assert counts_cf[eos] == 0 and counts_loop[eos] == 0
    tol = 4.0 * np.sqrt(analytic * (1 - analytic) / n) + 1e-4      # four standard deviations per token
    assert np.all(np.abs(counts_cf / n - analytic) < tol), (counts_cf / n, analytic)
    assert np.all(np.abs(counts_loop / n - analytic) < tol), (counts_loop / n, analytic)
    # and the "mask" policy is a different distribution (EOS gives up its nucleus slot): the reason both exist
    counts_m = np.zeros(v)
    for i in range(4000):
        counts_m[int(osyn.ras_sample(logits, hist, us[i], top_k, top_p, win, tau, eos, True, "mask")[0])] += 1
    assert np.abs(counts_m / 4000 - analytic).max() > 0.02


def test_kaldi_fbank_matches_the_kaldi_mimicking_feature_extractor():
    """astts.audio.kaldi_fbank / kaldi_mel_filterbank (the 80-bin Kaldi fbank the reference's speaker-embedding network takes: SURVEY.md a12)
    against transformers' SeamlessM4TFeatureExtractor (numpy path that mimics Kaldi: tests/golden/kaldi_fbank.npz, generated by
    make_synth_block_fixtures.py --kaldi; that extractor scales the samples by 2^15 as Kaldi's own tools do: scale=32768 here)."""
    from astts import audio

    fx = np.load(os.path.join(GOLD, "kaldi_fbank.npz"))
    fb = audio.kaldi_mel_filterbank(16000, 512, 80, 20.0, 0.0)
    assert fb.shape == (80, 257) and float(np.abs(fb.T - fx["mel_filters"]).max()) < 1e-7
    assert float(np.abs(np.power(np.hanning(400), 0.85) - fx["window"]).max()) < 1e-6
    wav = torch.from_numpy(fx["wav"])
    f = audio.kaldi_fbank(wav, scale=32768.0)[0].numpy()
    assert f.shape == fx["features"].shape == (1 + (wav.numel() - 400) // 160, 80)
    assert float(np.abs(f - fx["features"]).max()) < 2e-5
    # upstream's own scale (torchaudio takes the floats as they are): the same features shifted by 2 ln(32768) wherever the floor is not hit,
    # and the shift disappears with the mean over time that upstream subtracts
    g = audio.kaldi_fbank(wav, scale=1.0, subtract_mean=True)[0].numpy()
    h = f - f.mean(axis=0, keepdims=True)
    assert float(np.abs(g - h).max()) < 1e-4
    with pytest.raises(ValueError):
        audio.kaldi_fbank(torch.zeros(1, 399))
