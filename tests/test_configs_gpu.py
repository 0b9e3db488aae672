"""BASELINE configs 4 and 5 on the one GPU a test box has:

  config 4  one rank's shard of the IEMOCAP test set: ceil(1623 / 8) = 203 sentences, ragged, Ts_i = clamp(round(20 words_i),
            25, 1500) speech tokens, CosyVoice-300M widths, through the drop-in CosyVoice.inference_tts_with_st_batch surface
            (hot loop #2 of /root/reference/tts_with_rag.py:172-197, batched); retrieval of all 1623 queries against the
            1 000-row bank vs oracle/knn.py.
  config 3  64 long-form lines x 6 text segments (Tt = 64, forced Ts = 250: 30 s per line) through ITS driver,
            astts.cli.tts_with_style_and_timbre --batch_size 64 (/root/reference/tts_with_style_and_timbre.py:82-95), full model:
            64 wavs of 30 s; three sampled lines equal their one-at-a-time run.
  config 5  B = 256 utterances at the config-2 shapes in ONE engine call (rows 0..7 equal the batch-8 run: tokens bit for bit);
            the 100k x 6144 bank searched in three row shards + merged == the unsharded search == the oracle on a sample.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sentences():
    with open(os.path.join(ROOT, "tests", "golden", "iemocap_test_sentences.json")) as f:
        s = json.load(f)["all"]
    assert len(s) == 1623
    return s


def _ts_for(sentence: str) -> int:
    return int(min(max(round(20 * len(sentence.split())), 25), 1500))


def test_config4_knn_all_1623_queries_match_oracle():
    import bench
    from astts.knn import StyleBank
    from astts.parallel import shard_bounds
    from oracle import knn as oknn

    bank = bench.make_config2_bank(1000, 6144)
    q = bench.make_queries(bank, 1623, seed=4)
    sb = StyleBank(bank, device=DEV)
    idx, sc = sb.search(q, 3)
    ei, es = oknn.knn_search(bank, q, 3)
    assert np.array_equal(idx, ei)
    assert np.allclose(sc, es, atol=1e-6, rtol=0)
    # the rank shards of the data-parallel run (203 queries each, the last one short) return the same rows
    for r in (0, 7):
        b, e, per = shard_bounds(1623, 8, r)
        assert per == 203
        i_r, _ = sb.search(q[b:e], 3)
        assert np.array_equal(i_r, ei[b:e])
    print(f"config 4 retrieval: 1623 x 1000 x 6144, ids == oracle; {sb.last_fallbacks()} exact fallbacks in the last shard")


def test_config4_rank_shard_ragged_full_model():
    from astts.compat.cosyvoice import CosyVoice
    from astts.parallel import shard_bounds

    sents = _sentences()
    b0, e0, per = shard_bounds(len(sents), 8, 0)
    shard = sents[b0:e0]
    assert len(shard) == 203
    want = [_ts_for(s) for s in shard]
    cv = CosyVoice("/nonexistent", seed=0, allow_random_init=True)
    cfg = cv.cfg
    sr = 16000
    t = torch.arange(int(2.5 * sr)) / sr
    g = torch.Generator().manual_seed(0)
    style = (0.3 * torch.sin(2 * math.pi * 220 * t) + 0.01 * torch.randn(t.shape, generator=g))[None]
    timbre = (0.3 * torch.sin(2 * math.pi * 330 * t[: 2 * sr]) + 0.01 * torch.randn(2 * sr, generator=g))[None]
    style_text = "He did. In Niagara Falls."
    items = [(s, style_text, style, timbre) for s in shard]
    pm = cv.frontend.prompt(timbre).mel.shape[1]
    draws = [cv.make_draws(n, pm, 1000 + i) for i, n in enumerate(want)]
    outs = cv.inference_tts_with_st_batch(items, max_batch=32, split=False, fixed_tokens=want, draws=draws)
    toks, mels = cv.last_tokens, cv.last_mels
    assert len(outs) == 203 and all(len(o) == 1 for o in outs)
    capped = 0
    audio = 0.0
    for i, o in enumerate(outs):
        w = o[0]["tts_speech"]
        n_tok = int(toks[i].numel())
        capped += n_tok != want[i]
        assert n_tok <= want[i] and (n_tok == want[i] or n_tok >= 1000), (i, n_tok, want[i])     # only the position tables cap a row
        assert w.shape == (1, cfg.mel_frames_for_tokens(n_tok) * cfg.hop), (i, w.shape, n_tok)
        assert bool(torch.isfinite(w).all()) and float(w.abs().max()) <= cfg.audio_limit + 1e-6, i
        assert int(toks[i].max()) < cfg.speech_vocab and int(toks[i].min()) >= 0
        audio += w.shape[1] / cfg.sample_rate
    print(f"config 4 shard: 203 sentences, {sum(want)} tokens wanted, {capped} rows capped by the position tables, {audio:.0f} s of audio")
    # three sampled rows one at a time (a batch of one: no padding, no neighbours) with the same draws.  Free-running tokens: a
    # row's prefix comes out of other GEMM tiles in a 32-row batch (1 ulp), which sampling can amplify into a different token at
    # a near-tie, so the first tokens must agree and the agreement is printed; with the batch's tokens forced, the rendering of
    # the row alone must equal its rendering inside the ragged batch to the rounding noise of other tiles.
    order = sorted(range(203), key=lambda i: want[i])
    for i in (order[5], order[101], order[197]):
        one = cv.inference_tts_with_st_batch([items[i]], max_batch=1, split=False, fixed_tokens=[want[i]], draws=[draws[i]])
        t1 = cv.last_tokens[0]
        n = min(int(t1.numel()), int(toks[i].numel()))
        agree = float((t1[:n] == toks[i][:n]).float().mean())
        first = int((t1[:n] != toks[i][:n]).nonzero()[0]) if agree < 1.0 else n
        assert first >= min(25, n), (i, first)
        one_f = cv.inference_tts_with_st_batch([items[i]], max_batch=1, split=False, fixed_tokens=[want[i]], draws=[draws[i]],
                                               forced=[toks[i]])
        assert torch.equal(cv.last_tokens[0], toks[i])
        dm = float((cv.last_mels[0] - mels[i]).abs().max()) / float(mels[i].abs().max())
        wa, wb = one_f[0][0]["tts_speech"].double(), outs[i][0]["tts_speech"].double()
        snr = 10.0 * math.log10(float((wb ** 2).sum()) / max(float(((wa - wb) ** 2).sum()), 1e-30))
        # the vocoder runs per row, so the row's waveform inside the batch IS the vocoder on its batch mel: bit for bit
        d = draws[i]
        n_s = outs[i][0]["tts_speech"].shape[1]
        w_again = cv.engine.hift.forward(mels[i][None].to(DEV), d["phase0"].to(DEV), d["noise"][:, :n_s].to(DEV)).cpu()
        assert torch.equal(w_again, outs[i][0]["tts_speech"]), i
        print(f"config 4 row {i} ({want[i]} tokens): free-running agreement with the one-at-a-time run {agree:.3f} (first difference at {first}); "
              f"forced: mel rel diff {dm:.2e}, waveform SNR {snr:.1f} dB")
        # mel: other GEMM tiles in a 32-row ragged batch than in a batch of one (an order of magnitude inside the flow stage's
        # oracle tolerance).  Waveform: the harmonic source integrates the predicted f0, so a 4e-4 mel difference becomes a phase
        # drift that grows with the utterance (30 s rows: ~37 dB; 5 s rows: > 50 dB) -- bounded, but looser than the mel bar.
        assert dm < 1e-3 and snr > 30.0, i


def test_config3_longform_through_its_driver(tmp_path):
    """BASELINE config 3 through the drop-in driver: a text file of 64 lines, each 6 sentences of 64 byte-tokens (the stand-in
    tokenizer is one token per byte, so text_normalize cuts a line into exactly 6 segments of Tt = 64), one style and one timbre wav,
    `--batch_size 64 --fixed_tokens 250 --seed 3`: 384 segments in six ragged GPU batches of 64, the segments of a line
    concatenated (the documented divergence from the reference's overwrite, tts_with_style_and_timbre.py:94-95)."""
    import time

    from astts import audio
    from astts.cli import tts_with_style_and_timbre as drv
    from astts.compat.cosyvoice import CosyVoice, load_wav

    cv = CosyVoice("/nonexistent", seed=0, allow_random_init=True)
    cfg = cv.cfg
    sr = 16000
    t = torch.arange(3 * sr) / sr
    g = torch.Generator().manual_seed(0)
    audio.write_wav(str(tmp_path / "style_a.wav"), (0.3 * torch.sin(2 * math.pi * 220 * t) + 0.01 * torch.randn(t.shape, generator=g))[None], sr)
    audio.write_wav(str(tmp_path / "timbre_b.wav"), (0.3 * torch.sin(2 * math.pi * 330 * t) + 0.01 * torch.randn(t.shape, generator=g))[None], sr)
    words = ["alpha", "bravo", "delta", "gamma", "omega", "sigma", "theta", "kappa"]
    lines = []
    for i in range(64):
        sents = []
        for k in range(6):
            tag = " " + "abcdefghij"[i // 10] + "abcdefghij"[i % 10] + " " + "abcdefghij"[k]        # (letters: text_normalize spells digits out)
            body = " ".join(words[(i + k + j) % 8] for j in range(12))[:63 - len(tag)] + tag
            sents.append(body.ljust(63, "x") + ".")
        assert all(len(x.encode()) == 64 for x in sents)
        lines.append("".join(sents))          # (no blank between sentences: split_paragraph keeps a leading blank with its sentence)
    (tmp_path / "long.txt").write_text("\n".join(lines) + "\n", encoding="utf-8")
    from astts.frontend import text_normalize
    assert [len(cv.frontend.tokenizer.encode(x)) for x in text_normalize(lines[5], cv.frontend.tokenizer)] == [64] * 6
    style_text = "He did. In Niagara Falls."
    args = drv.build_parser().parse_args(["--style_wav_path", str(tmp_path / "style_a.wav"), "--timbre_wav_path", str(tmp_path / "timbre_b.wav"),
                                          "--style_wav_text", style_text, "--txt_path", str(tmp_path / "long.txt"),
                                          "--result_dir", str(tmp_path / "out"), "--batch_size", "64", "--fixed_tokens", "250", "--seed", "3"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    written = drv.tts_for_infer(args, cosyvoice=cv)
    dt = time.perf_counter() - t0
    assert [os.path.basename(p) for p in written] == [f"style_a_{i}_to_timbre_b.wav" for i in range(1, 65)]
    n_line = 6 * cfg.mel_frames_for_tokens(250) * cfg.hop
    secs = 0.0
    for p in written:
        w, rate = audio.read_wav(p)
        assert rate == 22050 and w.shape == (1, n_line) and bool(np.isfinite(w).all()) and float(np.abs(w).max()) <= cfg.audio_limit + 1e-6
        secs += w.shape[1] / rate
    assert secs >= 64 * 29.9
    print(f"config 3 through its driver: 64 lines x 6 segments, {secs:.0f} s of audio in {dt:.1f} s (prompts, host I/O and wav writing included) = {secs / dt:.0f}x real time")
    # three sampled lines one at a time (a batch of the line's own six segments) under the same per-line seed
    style_wav, timbre_wav = load_wav(str(tmp_path / "style_a.wav"), 16000), load_wav(str(tmp_path / "timbre_b.wav"), 16000)
    for i in (0, 31, 63):
        cnt = i + 1
        one = cv.inference_tts_with_st_batch([(lines[i], style_text, style_wav, timbre_wav)], max_batch=64, seeds=[3 * 1000003 + cnt], fixed_tokens=250)
        assert len(one[0]) == 6
        toks_one = [tk.clone() for tk in cv.last_tokens]
        w_one = torch.cat([j["tts_speech"] for j in one[0]], dim=1)
        w_drv = torch.from_numpy(audio.read_wav(written[i])[0])
        # free-running tokens of a 6-row batch vs the 64-row batch the driver formed: the prefix comes out of other GEMM tiles
        # (1 ulp), which sampling can amplify at a near-tie -- as in the config-4 test the first tokens must agree; when ALL tokens
        # agree the audio must too, to the rounding noise of other tiles
        same = all(int(tk.numel()) == 250 for tk in toks_one)
        assert same
        a, b = w_one.double(), w_drv.double()
        snr = 10.0 * math.log10(float((b ** 2).sum()) / max(float(((a - b) ** 2).sum()), 1e-30))
        print(f"config 3 line {cnt}: one-at-a-time vs driver waveform SNR {snr:.1f} dB")
        assert snr > 25.0, (cnt, snr)


def test_config5_batch256_rows_equal_batch8():
    from astts.synth.config import SynthConfig
    from astts.synth.model import SynthEngine
    from astts.synth.weights import make_all

    cfg = SynthConfig()
    eng = SynthEngine(make_all(cfg, 0), cfg, DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    B, Tt, Tp, Ts = 256, 32, 150, 250
    text = torch.randint(0, cfg.text_vocab, (B, Tt), device=DEV, generator=g)
    tlen = torch.full((B,), Tt, dtype=torch.int32, device=DEV)
    spk_s = torch.randn(B, cfg.spk_dim, device=DEV, generator=g)
    spk_t = torch.randn(B, cfg.spk_dim, device=DEV, generator=g)
    style_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=DEV, generator=g)
    timbre_tok = torch.randint(0, cfg.speech_vocab, (B, Tp), device=DEV, generator=g)
    tmp, tm = cfg.mel_frames_for_tokens(Tp), cfg.mel_frames_for_tokens(Ts)
    timbre_mel = torch.randn(B, tmp, cfg.mel, device=DEV, generator=g)
    u = torch.rand(Ts, B, 2, device=DEV, generator=g)
    z = torch.randn(B, tmp + tm, cfg.mel, device=DEV, generator=g)
    nh = cfg.nb_harmonics + 1
    phase0 = (torch.rand(B, nh, device=DEV, generator=g) * 2 - 1) * math.pi
    phase0[:, 0] = 0
    noise = torch.randn(B, tm * cfg.upsample_total, nh, device=DEV, generator=g)
    pre = eng.lm.prefix(text, tlen, spk_s, style_tok)
    toks = eng.lm.decode(pre, Ts, u, ignore_eos=True)                    # eight 32-row groups
    mel, wav = eng.tts_render(toks, timbre_tok, timbre_mel, spk_t, z, phase0, noise)
    torch.cuda.synchronize()
    assert toks.shape == (B, Ts) and wav.shape == (B, tm * cfg.upsample_total)
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= cfg.audio_limit + 1e-6
    assert int(toks.max()) < cfg.speech_vocab and int(toks.min()) >= 0
    # rows 0..7 as the benchmark's batch of 8 (same prefix VALUES: the text encoder picks its GEMM tile by row count)
    sl = slice(0, 8)
    def diff(a, b):
        d = (a != b)
        return f"{int(d.sum())} tokens differ in rows {sorted(set(d.nonzero()[:, 0].tolist()))}, first at step {int(d.nonzero()[:, 1].min()) if bool(d.any()) else -1}"

    toks8 = eng.lm.decode(pre[:, sl].contiguous(), Ts, u[:, sl].contiguous(), ignore_eos=True)
    assert torch.equal(toks8, toks[sl]), diff(toks8, toks[sl])            # the decode step is row-independent bit for bit at every width
    sl2 = slice(248, 256)
    toks8b = eng.lm.decode(pre[:, sl2].contiguous(), Ts, u[:, sl2].contiguous(), ignore_eos=True)
    assert torch.equal(toks8b, toks[sl2]), diff(toks8b, toks[sl2])
    mel8, wav8 = eng.tts_render(toks8, timbre_tok[sl], timbre_mel[sl], spk_t[sl], z[sl], phase0[sl], noise[sl])
    torch.cuda.synchronize()
    dm = float((mel8 - mel[sl]).abs().max()) / float(mel.abs().max())
    snr = 10.0 * math.log10(float((wav[sl].double() ** 2).sum()) / max(float(((wav8 - wav[sl]).double() ** 2).sum()), 1e-30))
    print(f"config 5: rows 0..7 of the 256-row batch vs the batch-8 run: tokens equal, mel rel diff {dm:.2e}, waveform SNR {snr:.1f} dB")
    assert dm < 1e-3 and snr > 40.0, (dm, snr)


def test_config5_bank_100k_x_6144_three_shards_equal_unsharded():
    from astts.knn import StyleBank
    from astts.parallel import bank_sharded_search, merge_topk, shard_bounds
    from oracle import knn as oknn

    n, d, nq, k = 100_000, 6144, 256, 3
    g = torch.Generator(device=DEV).manual_seed(1234)
    bank = torch.randn((n, d), generator=g, device=DEV, dtype=torch.float32).to(torch.float16)
    rows = torch.randint(0, n, (nq,), generator=g, device=DEV)
    q = bank[rows].to(torch.float32) + 0.7 * torch.randn((nq, d), generator=g, device=DEV)
    sb = StyleBank(bank)
    i_all, s_all, s64_all = (t.clone() for t in sb.search_device(q, k, return_f64=True))
    del sb
    parts = []
    for r in range(3):
        b, e, _ = shard_bounds(n, 3, r)
        sbr = StyleBank(bank[b:e])
        i_r, _, s_r = sbr.search_device(q, k, return_f64=True)
        parts.append((torch.where(i_r >= 0, i_r + b, i_r).clone(), s_r.clone()))
        # the one-rank form of the collective path as well (dist=None: merge of this shard alone)
        gi, gs = bank_sharded_search(lambda qq, kk: (i_r, s_r), q, k, b, None)
        assert torch.equal(gi, parts[-1][0])
        del sbr
    mi, ms = merge_topk(torch.cat([p[1] for p in parts], 1), torch.cat([p[0] for p in parts], 1), k)
    assert torch.equal(mi, i_all)
    assert torch.equal(ms, s64_all)                                       # fp64 cosines: the same bits from a shard as from the whole bank
    ei, es = oknn.knn_search(bank.cpu().numpy(), q[:16].cpu().numpy(), k)  # 16 queries x 100k x 6144 in fp64 on the host
    assert np.array_equal(mi[:16].cpu().numpy(), ei)
    assert np.allclose(ms[:16].cpu().numpy(), es, atol=1e-12, rtol=0)
    print("config 5: 100k x 6144 in 3 shards + merge == unsharded == oracle (16 sampled queries)")
