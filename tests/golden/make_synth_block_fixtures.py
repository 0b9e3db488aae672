"""Generates tests/golden/synth_blocks.npz: inputs, weights and expected outputs of the PUBLISHED building blocks that
oracle/synth.py restates, produced by independent third-party implementations of the same blocks in transformers
(the reference's own synthesis code is a private CosyVoice fork that exists nowhere in this environment:
/root/reference/tts_with_rag.py:1-2,18-19,159,195):

  rel-pos attention  FastSpeech2ConformerAttention + FastSpeech2ConformerRelPositionalEncoding and
                     Wav2Vec2ConformerSelfAttention("relative") + Wav2Vec2ConformerRelPositionalEmbedding
                     (espnet's RelPositionMultiHeadedAttention incl. pos_bias_u / pos_bias_v and the rel-shift)
                                                         -> oracle.synth.relpos_attention / rel_pos_table      (a13, a14)
  encoder layer      FastSpeech2ConformerEncoderLayer (pre-norm, no macaron / convolution module)
                                                         -> oracle.synth.relpos_layer (residual / norm order)  (a13, a14)
  Snake              dac Snake1d                         -> oracle.synth._snake                               (a15)
  HiFi-GAN resblock  speecht5 HifiGanResidualBlock       -> oracle.synth._resblock (structure: dilations, paddings, residuals)
  HiFi-GAN trunk     SpeechT5HifiGan (conv_pre, leaky-relu, ConvTranspose1d padding (k - u) / 2, mean over the parallel
                     resblocks, final leaky-relu 0.01)   -> oracle.synth.hift_trunk with a silent source path
  Whisper log-mel    WhisperFeatureExtractor(feature_size=128) -> astts.audio.whisper_log_mel / mel_filterbank (a12)
  prompt log-mel     transformers.audio_utils.mel_filter_bank / spectrogram -> astts.audio.mel_filterbank / mel_spectrogram (a12 / a14)
  transformer block  torch.nn.TransformerEncoderLayer(norm_first=True, gelu) -> oracle.synth._tfm_block (the estimator's BasicTransformerBlock) (a14)
  nucleus set        TopPLogitsWarper (generation/logits_process.py) -> oracle.synth.nucleus (the sampler's candidate set)  (a13)
  Kaldi fbank        SeamlessM4TFeatureExtractor._extract_fbank_features (numpy "mimic Kaldi" path)
                                                         -> astts.audio.kaldi_fbank / kaldi_mel_filterbank  (a12; own file kaldi_fbank.npz)
  CFM t-grid + CFG   Qwen2_5OmniToken2WavDiTModel.sample (sway_coefficient -1 = the cosine grid; guided + (guided - null) g)
                                                         -> oracle.synth.cfm_t_grid / cfg_combine  (a14; own file cfm_grid_cfg.npz)

Run in the BUILD container only (python tests/golden/make_synth_block_fixtures.py).  The .npz is data: seeded random
weights and inputs in, the third-party outputs out.  Nothing of transformers travels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "autostyle-tts_amd")]


def t2n(sd, prefix):
    return {prefix + k: v.detach().float().numpy() for k, v in sd.items()}


def relpos_fastspeech2(out):
    from transformers import FastSpeech2ConformerConfig
    from transformers.models.fastspeech2_conformer import modeling_fastspeech2_conformer as m

    torch.manual_seed(11)
    d, heads, b, t = 96, 4, 3, 29
    cfg = FastSpeech2ConformerConfig(hidden_size=d)
    mc = {"num_attention_heads": heads, "attention_dropout_rate": 0.0, "positional_dropout_rate": 0.0}
    att = m.FastSpeech2ConformerAttention(cfg, mc).eval()
    pos = m.FastSpeech2ConformerRelPositionalEncoding(cfg, mc).eval()
    with torch.no_grad():
        for p in att.parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() > 1 else 0.2))
        x = torch.randn(b, t, d)
        lens = torch.tensor([t, t - 6, 11])
        mask = (torch.arange(t)[None, :] < lens[:, None]).long()[:, None, :]           # (batch, 1, time)
        _, pos_emb = pos(x)
        y, _ = att(x, attention_mask=mask, pos_emb=pos_emb)
    out.update(t2n(att.state_dict(), "fs2."))
    out.update({"fs2.x": x.numpy(), "fs2.lens": lens.numpy(), "fs2.pos_emb": pos_emb.numpy(), "fs2.y": y.numpy(),
                "fs2.heads": np.int64(heads)})
    print("fastspeech2 rel-pos attention:", tuple(y.shape), "pos_emb", tuple(pos_emb.shape))


def encoder_layer(out):
    """Pre-norm conformer layer without the macaron feed-forward and the convolution module = the layer of espnet's
    TransformerEncoder with rel-pos attention (what CosyVoice's text encoder, token encoder and LM body stack)."""
    from transformers import FastSpeech2ConformerConfig
    from transformers.models.fastspeech2_conformer import modeling_fastspeech2_conformer as m

    torch.manual_seed(16)
    d, heads, b, t, ffn = 64, 2, 2, 21, 160
    cfg = FastSpeech2ConformerConfig(hidden_size=d, use_macaron_style_in_conformer=False, use_cnn_in_conformer=False,
                                     positionwise_conv_kernel_size=1)
    mc = {"num_attention_heads": heads, "attention_dropout_rate": 0.0, "positional_dropout_rate": 0.0, "dropout_rate": 0.0,
          "linear_units": ffn, "normalize_before": True, "concat_after": False}
    lay = m.FastSpeech2ConformerEncoderLayer(cfg, mc).eval()
    lay.conv_module = None            # transformers 5.15 reads the attribute even when the module is configured off
    pos = m.FastSpeech2ConformerRelPositionalEncoding(cfg, mc).eval()
    with torch.no_grad():
        for n_, p in lay.named_parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() > 1 else 0.2) + (1.0 if n_.endswith("layer_norm.weight") else 0.0))
        x = torch.randn(b, t, d)
        lens = torch.tensor([t, 13])
        mask = (torch.arange(t)[None, :] < lens[:, None]).long()[:, None, :]
        _, pos_emb = pos(x)
        y = lay(x, pos_emb=pos_emb, attention_mask=mask)[0]
    out.update(t2n(lay.state_dict(), "enc."))
    out.update({"enc.x": x.numpy(), "enc.lens": lens.numpy(), "enc.y": y.numpy(), "enc.heads": np.int64(heads)})
    print("conformer encoder layer:", tuple(y.shape), sorted(k for k in lay.state_dict())[:4], "...")


def relpos_wav2vec2(out):
    from transformers import Wav2Vec2ConformerConfig
    from transformers.models.wav2vec2_conformer import modeling_wav2vec2_conformer as w

    torch.manual_seed(12)
    d, heads, b, t = 64, 2, 2, 37
    cfg = Wav2Vec2ConformerConfig(hidden_size=d, num_attention_heads=heads, position_embeddings_type="relative",
                                  attention_dropout=0.0, max_source_positions=100)
    att = w.Wav2Vec2ConformerSelfAttention(cfg).eval()
    pos = w.Wav2Vec2ConformerRelPositionalEmbedding(cfg).eval()
    with torch.no_grad():
        for p in att.parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() > 1 else 0.2))
        x = torch.randn(b, t, d)
        rel = pos(x)
        y, _ = att(x, attention_mask=None, relative_position_embeddings=rel)
    out.update(t2n(att.state_dict(), "w2v."))
    out.update({"w2v.x": x.numpy(), "w2v.pos_emb": rel.numpy(), "w2v.y": y.numpy(), "w2v.heads": np.int64(heads)})
    print("wav2vec2-conformer rel-pos attention:", tuple(y.shape))


def snake(out):
    from transformers.models.dac.modeling_dac import Snake1d

    torch.manual_seed(13)
    s = Snake1d(24).eval()
    with torch.no_grad():
        s.alpha.copy_(torch.rand_like(s.alpha) * 3 + 0.05)
        x = torch.randn(2, 24, 50) * 2
        y = s(x)
    out.update({"snake.alpha": s.alpha.detach().view(-1).numpy(), "snake.x": x.numpy(), "snake.y": y.numpy()})


def hifigan(out):
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
    from transformers.models.speecht5.modeling_speecht5 import HifiGanResidualBlock

    torch.manual_seed(14)
    # one residual block at each kernel size HiFT uses
    for k in (3, 7, 11):
        rb = HifiGanResidualBlock(12, k, (1, 3, 5), 0.1).eval()
        with torch.no_grad():
            for p in rb.parameters():
                p.copy_(torch.randn_like(p) * 0.2)
            x = torch.randn(2, 12, 90)
            y = rb(x)
        out.update(t2n(rb.state_dict(), f"rb{k}."))
        out.update({f"rb{k}.x": x.numpy(), f"rb{k}.y": y.numpy()})
    # the generator trunk: conv_pre -> 2 x [leaky-relu -> ConvTranspose1d(16, stride 8, padding 4) -> mean of 3 resblocks] -> leaky-relu
    cfg = SpeechT5HifiGanConfig(model_in_dim=10, upsample_initial_channel=32, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16],
                                resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1,
                                normalize_before=False)
    g = SpeechT5HifiGan(cfg).eval()
    grabbed = {}
    g.conv_post.register_forward_pre_hook(lambda mod, args: grabbed.__setitem__("pre_post", args[0].detach().clone()))
    with torch.no_grad():
        for p in g.parameters():
            p.copy_(torch.randn_like(p) * 0.15)
        mel = torch.randn(2, 6, 10)                         # (batch, frames, mel)
        wav = g(mel)
    sd = {k: v for k, v in g.state_dict().items() if k not in ("mean", "scale")}
    out.update(t2n(sd, "gan."))
    out.update({"gan.mel": mel.numpy(), "gan.pre_post": grabbed["pre_post"].numpy(), "gan.wav": wav.numpy()})
    print("hifigan trunk:", tuple(grabbed["pre_post"].shape), "->", tuple(wav.shape))


def whisper(out):
    from transformers import WhisperFeatureExtractor

    fe = WhisperFeatureExtractor(feature_size=128)
    g = torch.Generator().manual_seed(15)
    n = 16000 * 2 + 123
    t = torch.arange(n) / 16000.0
    wav = (0.4 * torch.sin(2 * np.pi * 220 * t) + 0.2 * torch.sin(2 * np.pi * 1870 * t + 1.0) + 0.05 * torch.randn(n, generator=g)).numpy()
    feats = fe(wav, sampling_rate=16000, return_tensors="np", padding="do_not_pad")["input_features"][0]     # [128, frames]
    out.update({"whisper.wav": wav.astype(np.float32), "whisper.features": feats.astype(np.float32),
                "whisper.mel_filters": np.asarray(fe.mel_filters, np.float32)})                              # [201, 128]
    print("whisper log-mel:", feats.shape)


def prenorm_transformer_block(out):
    """torch.nn.TransformerEncoderLayer(norm_first=True, activation="gelu") = the arithmetic of diffusers' BasicTransformerBlock as
    the Matcha / CosyVoice estimator configures it (self-attention without q/k/v bias, out-projection with bias, exact-erf GELU
    feed-forward, pre-LayerNorm, two residuals) -> oracle.synth._tfm_block."""
    torch.manual_seed(18)
    d, heads, ffn, b, t = 128, 2, 512, 3, 37
    lay = torch.nn.TransformerEncoderLayer(d, heads, ffn, dropout=0.0, activation="gelu", batch_first=True, norm_first=True).eval()
    with torch.no_grad():
        for n_, p in lay.named_parameters():
            p.copy_(torch.randn_like(p) * (0.15 if p.dim() > 1 else 0.2) + (1.0 if n_.startswith("norm") and n_.endswith("weight") else 0.0))
        lay.self_attn.in_proj_bias.zero_()
        x = torch.randn(b, t, d)
        lens = torch.tensor([t, 20, 9])
        pad = torch.arange(t)[None, :] >= lens[:, None]
        y = lay(x, src_key_padding_mask=pad)
    sd = lay.state_dict()
    wq, wk, wv = sd["self_attn.in_proj_weight"].chunk(3, 0)
    out.update({"tfm.x": x.numpy(), "tfm.lens": lens.numpy(), "tfm.y": y.detach().numpy(), "tfm.heads": np.int64(heads),
                "tfm.norm1.weight": sd["norm1.weight"].numpy(), "tfm.norm1.bias": sd["norm1.bias"].numpy(),
                "tfm.attn1.to_q.weight": wq.numpy(), "tfm.attn1.to_k.weight": wk.numpy(), "tfm.attn1.to_v.weight": wv.numpy(),
                "tfm.attn1.to_out.0.weight": sd["self_attn.out_proj.weight"].numpy(), "tfm.attn1.to_out.0.bias": sd["self_attn.out_proj.bias"].numpy(),
                "tfm.norm3.weight": sd["norm2.weight"].numpy(), "tfm.norm3.bias": sd["norm2.bias"].numpy(),
                "tfm.ff.net.0.proj.weight": sd["linear1.weight"].numpy(), "tfm.ff.net.0.proj.bias": sd["linear1.bias"].numpy(),
                "tfm.ff.net.2.weight": sd["linear2.weight"].numpy(), "tfm.ff.net.2.bias": sd["linear2.bias"].numpy()})
    print("pre-norm transformer block:", tuple(y.shape))


def prompt_mel(out):
    """transformers.audio_utils (mel_filter_bank norm / scale "slaney", spectrogram with power 1, natural log, floor 1e-5) on the
    matcha-style prompt mel's parameters: 22 050 Hz, n_fft = win = 1024, hop 256, 80 bins over 0-8 kHz, reflect padding of
    (n_fft - hop) / 2 and no centring -> astts.audio.mel_filterbank / mel_spectrogram (the flow's prompt features)."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

    fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0, sampling_rate=22050,
                         norm="slaney", mel_scale="slaney")
    g = torch.Generator().manual_seed(19)
    n = 16000
    t = torch.arange(n) / 22050.0
    wav = (0.4 * torch.sin(2 * np.pi * 330 * t) + 0.15 * torch.sin(2 * np.pi * 2750 * t + 0.5) + 0.1 * torch.randn(n, generator=g)).numpy().astype(np.float32)
    pad = (1024 - 256) // 2
    y = np.pad(wav.astype(np.float64), (pad, pad), mode="reflect")
    s = spectrogram(y, window_function(1024, "hann", periodic=True), frame_length=1024, hop_length=256, fft_length=1024, power=1.0,
                    center=False, mel_filters=fb, mel_floor=1e-5, log_mel="log")
    out.update({"pmel.wav": wav, "pmel.filters": fb.T.astype(np.float32), "pmel.logmel": s.T.astype(np.float32)})
    print("prompt mel:", s.T.shape)


def nucleus_sets(out):
    """Top-p candidate sets of transformers' TopPLogitsWarper on seeded logits of three shapes (flat, peaked, with exact ties):
    upstream's nucleus_sampling adds tokens in descending probability while the mass already added is < top_p -- the smallest
    prefix reaching top_p, which is what the warper keeps -- and stops at top_k entries."""
    from transformers.generation.logits_process import TopPLogitsWarper

    g = torch.Generator().manual_seed(17)
    v = 4097
    flat = torch.randn(6, v, generator=g) * 0.7
    peaked = torch.randn(6, v, generator=g) * 4.0
    tied = torch.round(torch.randn(4, v, generator=g) * 3.0) / 2.0          # many exactly equal logits
    logits = torch.cat([flat, peaked, tied], 0)
    kept = {}
    for top_p in (0.8, 0.5, 0.95):
        w = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)
        sc = w(None, logits.clone())
        kept[top_p] = torch.isfinite(sc).numpy()
    out.update({"nucleus.logits": logits.numpy(), "nucleus.kept_p80": kept[0.8], "nucleus.kept_p50": kept[0.5], "nucleus.kept_p95": kept[0.95]})
    print("top-p sets:", {k: v_.sum(1).tolist() for k, v_ in kept.items()})


def kaldi_fbank_fixture():
    """80-bin Kaldi fbank of a seeded waveform by transformers' SeamlessM4TFeatureExtractor._extract_fbank_features (its numpy
    "mimic Kaldi" path: povey window, pre-emphasis 0.97, DC removal, 512-point power spectrum, Kaldi mel bank, log with float32-eps
    floor, samples scaled by 2^15) -> tests/golden/kaldi_fbank.npz.  Held against astts.audio.kaldi_fbank(scale=32768)."""
    from transformers import SeamlessM4TFeatureExtractor

    fe = SeamlessM4TFeatureExtractor(feature_size=80, num_mel_bins=80, sampling_rate=16000)
    g = torch.Generator().manual_seed(31)
    n = 16000 + 777
    t = torch.arange(n) / 16000.0
    wav = (0.3 * torch.sin(2 * np.pi * 180 * t) + 0.1 * torch.sin(2 * np.pi * 2310 * t + 0.5) + 0.02 * torch.randn(n, generator=g) + 0.01).numpy()
    feats = fe._extract_fbank_features(wav.astype(np.float64))                         # [frames, 80]
    path = os.path.join(ROOT, "tests", "golden", "kaldi_fbank.npz")
    np.savez_compressed(path, wav=wav.astype(np.float32), features=feats.astype(np.float32), mel_filters=np.asarray(fe.mel_filters, np.float32),
                        window=np.asarray(fe.window, np.float32))
    print("kaldi fbank:", feats.shape, "->", path, os.path.getsize(path) // 1024, "KB")


def cfm_grid_and_cfg():
    """The time grid and the classifier-free-guidance combination of a flow-matching sampler, as transformers'
    Qwen2_5OmniToken2WavDiTModel.sample codes them (models/qwen2_5_omni/modeling_qwen2_5_omni.py: `time_embedding += sway * (cos(pi/2 t)
    - 1 + t)`, `guided + (guided - null) * guidance_scale`).  `sample` is run UNBOUND on a stand-in model whose forward is a seeded
    closed-form map (cond half | uncond half), with the module's ODE solver replaced by a recorder: what is recorded is the grid the
    solver is handed and the guided velocity `sample`'s own ode_function returns for probe states.  -> tests/golden/cfm_grid_cfg.npz"""
    from transformers.models.qwen2_5_omni import modeling_qwen2_5_omni as m

    g = torch.Generator().manual_seed(23)
    mel, b, t_code, repeats = 12, 3, 9, 2
    a_c = torch.randn(mel, mel, generator=g) * 0.4
    a_u = torch.randn(mel, mel, generator=g) * 0.4
    rec = {"dc": [], "du": [], "x": [], "t": [], "guided": []}

    class StandIn:
        mel_dim = mel

        class config:
            max_position_embeddings = 1 << 20

        def __init__(self):
            self.repeats = repeats

        def __call__(self, hidden_states, quantized_code, speaker_embedding, condition_vector, time_step, apply_cfg=True, **kw):
            assert apply_cfg
            d_c = torch.tanh(hidden_states @ a_c + time_step) + 0.1 * condition_vector.mean()
            d_u = torch.sin(hidden_states @ a_u - time_step)
            rec["dc"].append(d_c.clone()); rec["du"].append(d_u.clone()); rec["x"].append(hidden_states.clone())
            rec["t"].append(torch.as_tensor(time_step).clone())
            return torch.cat([d_c, d_u], dim=0)

    class Recorder:
        def __init__(self, function, initial_value):
            self.function, self.initial_value = function, initial_value

        def integrate(self, time_points):
            rec["grid"] = time_points.clone()
            x = self.initial_value
            for i, t in enumerate(time_points[:-1]):
                v = self.function(t, x)
                rec["guided"].append(v.clone())
                x = x + (time_points[i + 1] - t) * v          # any trajectory will do: the probes only have to differ
            return torch.stack([x] * len(time_points))

    solver = m.RungeKutta4ODESolver
    m.RungeKutta4ODESolver = Recorder
    try:
        torch.manual_seed(5)
        out = {}
        for name, steps, scale in (("a", 11, 0.7), ("b", 6, 0.5)):
            for k in rec:
                rec[k] = [] if k != "grid" else None
            m.Qwen2_5OmniToken2WavDiTModel.sample(StandIn(), conditioning_vector=torch.randn(b, 7, generator=g),
                                                  reference_mel_spectrogram=torch.randn(b, t_code * repeats, mel, generator=g),
                                                  quantized_code=torch.zeros(b, t_code, dtype=torch.long), num_steps=steps,
                                                  guidance_scale=scale, sway_coefficient=-1.0)
            out.update({f"{name}.grid": rec["grid"].numpy(), f"{name}.scale": np.float64(scale), f"{name}.dc": torch.stack(rec["dc"]).numpy(),
                        f"{name}.du": torch.stack(rec["du"]).numpy(), f"{name}.guided": torch.stack(rec["guided"]).numpy()})
            print("cfm grid", name, rec["grid"].numpy().round(4).tolist())
    finally:
        m.RungeKutta4ODESolver = solver
    path = os.path.join(ROOT, "tests", "golden", "cfm_grid_cfg.npz")
    np.savez_compressed(path, **out)
    print("->", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    if "--cfm" in sys.argv:
        cfm_grid_and_cfg()
        sys.exit(0)
    if "--kaldi" in sys.argv:
        kaldi_fbank_fixture()
        sys.exit(0)
    fx = {}
    relpos_fastspeech2(fx)
    relpos_wav2vec2(fx)
    encoder_layer(fx)
    snake(fx)
    hifigan(fx)
    whisper(fx)
    nucleus_sets(fx)
    prompt_mel(fx)
    prenorm_transformer_block(fx)
    path = os.path.join(ROOT, "tests", "golden", "synth_blocks.npz")
    np.savez_compressed(path, **fx)
    print("->", path, os.path.getsize(path) // 1024, "KB,", len(fx), "arrays")
