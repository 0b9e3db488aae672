"""GPU parity of the learned frontend (astts/frontend_nets.py: speech tokenizer + CAM++ speaker network on HIP kernels) against
oracle/frontend_nets.py (fp32 torch on the CPU) under identical seeded weights (astts.frontend_weights).

Tolerances (fp16 weights / MFMA operands, fp32 accumulation and residual streams, against all-fp32): encoder frames and
frame-level speaker features <= 3e-3 of the tensor's scale, the 192-d embedding <= 5e-3 (observed values are printed).  Speech
tokens are an arg-min: on the frames THIS path computed they are bit-exact (the quantiser is a certified fp64 search); against the
oracle's own frames a token may differ only where the oracle's two nearest codes are closer than the frame error can decide."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def test_glue_operators_against_their_definitions():
    from astts import frontend_nets as fn

    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 200, generator=g)
    sc, sh = torch.rand(160, generator=g) + 0.5, torch.randn(160, generator=g)
    y = fn.affine_act(x.to(DEV), sc.to(DEV), sh.to(DEV), relu=True, out_dtype=torch.float32, cols=160).cpu()
    assert torch.allclose(y, torch.relu(x[:, :160] * sc + sh), atol=1e-6)
    y16 = fn.affine_act(x.to(DEV), sc.to(DEV), sh.to(DEV), relu=False, out_dtype=torch.float16, cols=160).cpu()
    assert torch.equal(y16, (x[:, :160] * sc + sh).to(torch.float16)) or torch.allclose(y16.float(), x[:, :160] * sc + sh, atol=4e-3)
    assert torch.equal(fn.affine_act(x.to(DEV), None, None, relu=True, out_dtype=torch.float32).cpu(), torch.relu(x))
    # frequency unfold: the rows f * sf + kf - 1 of a window side by side, zero outside
    z = torch.randn(2, 9, 5, 4, generator=g)
    for sf, nkf in ((1, 3), (2, 3), (2, 1)):
        f_out = (9 + 2 * ((nkf - 1) // 2) - nkf) // sf + 1
        u = fn.freq_unfold(z.to(DEV), f_out, sf, nkf).cpu().float()
        pad = torch.nn.functional.pad(z, (0, 0, 0, 0, (nkf - 1) // 2, (nkf - 1) // 2))
        want = torch.stack([torch.cat([pad[:, fo * sf + kf] for kf in range(nkf)], dim=-1) for fo in range(f_out)], dim=1)
        assert torch.equal(u, want.to(torch.float16).float()), (sf, nkf)
    assert torch.equal(fn.ftc_to_tfc(z.to(DEV)).cpu(), z.permute(0, 2, 1, 3).reshape(2, 5, 36))
    # context: mean over time + mean of the frame's segment (the last segment shorter)
    h = torch.randn(3, 250, 128, generator=g)
    ctx = fn.cam_context(h.to(DEV), 100).cpu()
    want = torch.stack([h[:, s * 100:(s + 1) * 100].mean(1) for s in range(3)], 1) + h.mean(1, keepdim=True)
    assert ctx.shape == (3, 3, 128) and torch.allclose(ctx, want, atol=2e-6)
    ctx16 = fn.cam_context(h.to(DEV).to(torch.float16), 100).cpu()
    assert torch.allclose(ctx16, torch.stack([h.half().float()[:, s * 100:(s + 1) * 100].mean(1) for s in range(3)], 1) + h.half().float().mean(1, keepdim=True), atol=2e-6)
    yv, m = torch.randn(3, 250, 32, generator=g), torch.randn(3, 3, 32, generator=g)
    buf = torch.zeros(3, 250, 96, device=DEV)
    fn.cam_gate(yv.to(DEV), m.to(DEV), buf[:, :, 32:64], 100)
    want = yv * torch.sigmoid(m.repeat_interleave(100, dim=1)[:, :250])
    assert torch.allclose(buf[:, :, 32:64].cpu(), want, atol=1e-6) and float(buf[:, :, :32].abs().max()) == 0 and float(buf[:, :, 64:].abs().max()) == 0
    sp = fn.stats_pool((h * 3 + 100).to(DEV)).cpu()
    assert torch.allclose(sp[:, :128], (h * 3 + 100).mean(1), atol=1e-4) and torch.allclose(sp[:, 128:], (h * 3 + 100).std(1, unbiased=True), rtol=1e-4)
    n = fn.l2_normalize(x.to(DEV)).cpu()
    assert torch.allclose(n, x / x.norm(dim=1, keepdim=True), atol=1e-6)
    assert torch.equal(fn.l2_normalize(torch.zeros(2, 8, device=DEV)).cpu(), torch.zeros(2, 8))


@pytest.mark.parametrize("shape,frames", [("tiny", 180), ("full", 300), ("full", 1502), ("full", 3000), ("full", 21)])
def test_speech_tokenizer_matches_oracle(shape, frames):
    """``frames`` log-mel frames (100 Hz): 3 s, 15 s, the 30 s maximum (all 1500 encoder positions) and a 0.2 s prompt at the published
    widths (d 1280, 20 heads, 6 blocks, 4096 codes)."""
    from astts.frontend_nets import SpeechTokenizerV1
    from astts.frontend_weights import SpeechTokenizerShape, make_speech_tokenizer_weights
    from oracle import frontend_nets as ofn

    cfg = SpeechTokenizerShape.tiny() if shape == "tiny" else SpeechTokenizerShape()
    sd = make_speech_tokenizer_weights(cfg, 17)
    tok = SpeechTokenizerV1(sd, cfg, DEV)
    g = torch.Generator().manual_seed(frames)
    mel = torch.randn(1, cfg.n_mels, frames, generator=g) * 0.5
    x = tok.encode(mel.to(DEV))
    ref, lens = ofn.tokenizer_encode(sd, cfg, mel)
    assert x.shape == (int(lens[0]), cfg.d)
    err = _rel(x, ref[0])
    print(f"[parity] speech tokenizer {shape} T={frames}: encoder frames rel err {err:.2e}")
    assert err < 3e-3
    codes = tok.quantize(x).cpu().long()
    own = ofn.vq_encode(sd, cfg, x.cpu())                                  # the oracle's arg-min on THIS path's frames
    assert torch.equal(codes, own)                                         # bit-exact: a certified fp64 search
    assert tok.codebook.last_fallbacks() <= x.shape[0]
    ref_codes = ofn.vq_encode(sd, cfg, ref[0])
    diff = (codes != ref_codes).nonzero().view(-1)
    agree = 1.0 - diff.numel() / codes.numel()
    print(f"[parity] speech tokenizer {shape} T={frames}: {agree:.4f} of the tokens equal the oracle's")
    # a differing token is a near-tie of the ORACLE: its two best codes closer than the frame error moves a distance
    e = sd["quantizer._codebook.embed"].double()
    for i in diff.tolist():
        f = ref[0, i].double()
        f = f / f.norm()
        d2 = ((e - f) ** 2).sum(1)
        best2 = torch.topk(-d2, 2).values.neg()
        assert float(best2[1] - best2[0]) < 4 * 3e-3, (i, best2)           # |d^2(a) - d^2(b)| moves by <= 2 |f' - f| |a - b| <= 4 x frame error
        assert int(codes[i]) in torch.topk(-d2, 4).indices.tolist()
    assert agree > 0.9


@pytest.mark.parametrize("shape,frames,batch", [("tiny", 130, 2), ("full", 298, 1), ("full", 1498, 2), ("full", 2998, 1), ("full", 18, 3)])
def test_campplus_matches_oracle(shape, frames, batch):
    """3 s, 15 s, 30 s and 0.2 s prompts (10 ms fbank frames; the last one shorter than a context segment) at the published CAM++ shape;
    stage by stage, then the embedding."""
    from astts.frontend_nets import CamPlusSpeakerNet
    from astts.frontend_weights import CamPlusShape, make_campplus_weights
    from oracle import frontend_nets as ofn

    cfg = CamPlusShape.tiny() if shape == "tiny" else CamPlusShape()
    sd = make_campplus_weights(cfg, 23)
    net = CamPlusSpeakerNet(sd, cfg, DEV)
    g = torch.Generator().manual_seed(frames)
    fb = torch.randn(batch, frames, cfg.feat_dim, generator=g) * 2.0
    fb = fb - fb.mean(dim=1, keepdim=True)
    head = net.head(fb.to(DEV))
    ref_head = ofn.campplus_head(sd, cfg, fb)                               # [B, c * F/8 + f, T]
    m, fo = cfg.m_channels, cfg.feat_dim // 8
    ref_head_tfc = ref_head.view(batch, m, fo, frames).permute(0, 3, 2, 1).reshape(batch, frames, fo * m)
    e_h = _rel(head, ref_head_tfc)
    fr = net.frames(head)
    ref_fr = ofn.campplus_xvector(sd, cfg, ref_head, return_frames=True).transpose(1, 2)
    e_f = _rel(fr, ref_fr)
    emb = net.embed(fb.to(DEV))
    ref_emb = ofn.speaker_embedding(sd, cfg, fb)
    e_e = _rel(emb, ref_emb)
    print(f"[parity] CAM++ {shape} T={frames} B={batch}: head {e_h:.2e}, frame features {e_f:.2e}, embedding {e_e:.2e}")
    assert emb.shape == (batch, cfg.emb)
    assert e_h < 3e-3 and e_f < 3e-3 and e_e < 5e-3
    if batch > 1:                                                           # rows do not depend on the batch they run in
        alone = net.embed(fb[1:].to(DEV))
        assert torch.equal(alone, emb[1:])


def test_from_waveform_end_to_end():
    """The two callables the Frontend holds: 16 kHz waveform -> tokens at 50 Hz / a 192-d vector, through the HIP feature kernels
    (Whisper log-mel, Kaldi fbank) -- against the oracle fed the host-side features of the same waveform."""
    from astts import audio
    from astts.frontend_nets import CamPlusSpeakerNet, SpeechTokenizerV1
    from astts.frontend_weights import CamPlusShape, SpeechTokenizerShape, make_campplus_weights, make_speech_tokenizer_weights
    from oracle import frontend_nets as ofn

    g = torch.Generator().manual_seed(5)
    n = 16000 * 4 + 123
    t = torch.arange(n) / 16000.0
    wav = (0.3 * torch.sin(2 * np.pi * 220 * t) + 0.2 * torch.sin(2 * np.pi * 1333 * t + 1.0) + 0.05 * torch.randn(n, generator=g))[None, :]
    tc, cc = SpeechTokenizerShape(), CamPlusShape()
    tsd, csd = make_speech_tokenizer_weights(tc, 1), make_campplus_weights(cc, 2)
    tok, net = SpeechTokenizerV1(tsd, tc, DEV), CamPlusSpeakerNet(csd, cc, DEV)
    toks = tok(wav)
    assert toks.dtype == torch.int32 and toks.shape == (1, (n // 160 - 1) // 2 + 1) and int(toks.min()) >= 0 and int(toks.max()) < tc.codes
    ref_codes, _ = ofn.speech_tokens(tsd, tc, audio.whisper_log_mel(wav))
    agree = float((toks[0].long() == ref_codes[0]).float().mean())
    print(f"[parity] speech tokens from a waveform: {agree:.4f} equal the oracle's")
    assert agree > 0.9
    emb = net(wav)
    ref = ofn.speaker_embedding(csd, cc, audio.kaldi_fbank(wav, n_mels=80, subtract_mean=True))
    e = _rel(emb, ref)
    print(f"[parity] speaker embedding from a waveform: rel err {e:.2e}")
    assert emb.shape == (1, 192) and e < 5e-3
    with pytest.raises(ValueError):
        tok(torch.zeros(1, 30 * 16000 + 1))
