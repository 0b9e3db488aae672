"""GPU end-to-end tests of the drop-in call surface: the reference's driver flows (tts_with_rag.py,
tts_with_style_and_timbre.py, search_embeddings.py) run through the CosyVoice / MilvusClient shims on the
MI355X engine with a tiny random-init model and wav files created on the fly."""
import json
import os
from datetime import datetime

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tone(path, seconds, f, sr=16000):
    from astts import audio

    t = torch.arange(int(seconds * sr)) / sr
    audio.write_wav(path, (0.3 * torch.sin(2 * np.pi * f * t) + 0.01 * torch.randn(t.shape))[None, :], sr)


@pytest.fixture(scope="module")
def cosy():
    from astts.compat.cosyvoice import CosyVoice
    from astts.synth.config import SynthConfig

    with pytest.warns(UserWarning, match="RANDOM-INIT"):
        cv = CosyVoice("/nonexistent/CosyVoice-300M", config=SynthConfig.tiny(), seed=0, allow_random_init=True)
    assert cv.random_init
    return cv


def test_inference_generators_contract(cosy, tmp_path):
    from astts.compat.cosyvoice import load_wav

    _tone(str(tmp_path / "style.wav"), 1.5, 220.0, sr=22050)      # load_wav resamples to 16 kHz
    _tone(str(tmp_path / "timbre.wav"), 1.2, 330.0)
    style = load_wav(str(tmp_path / "style.wav"), 16000)
    timbre = load_wav(str(tmp_path / "timbre.wav"), 16000)
    assert style.shape[0] == 1 and abs(style.shape[1] - 24000) <= 1
    gen = cosy.inference_tts_with_st("I did it, I asked her to marry me.", "He did. In Niagara Falls.", style, timbre, stream=False)
    assert hasattr(gen, "__next__")                                # lazy generator, as upstream
    outs = list(gen)
    assert len(outs) == 1
    w = outs[0]["tts_speech"]
    assert w.dtype == torch.float32 and w.device.type == "cpu" and w.dim() == 2 and w.shape[0] == 1
    assert w.shape[1] % cosy.cfg.upsample_total == 0 and w.shape[1] > 0
    assert torch.isfinite(w).all() and float(w.abs().max()) <= 0.99 + 1e-6
    # token count honours the 2x..20x text-length window of the LM (EOS masked below, capped above)
    n_text = len("I did it, I asked her to marry me.".encode())
    n_frames = w.shape[1] // cosy.cfg.upsample_total
    assert cosy.cfg.mel_frames_for_tokens(2 * n_text) <= n_frames <= cosy.cfg.mel_frames_for_tokens(20 * n_text)
    zs = list(cosy.inference_zero_shot("Guess what?", "I do. Yeah.", style, stream=False))
    assert len(zs) == 1 and zs[0]["tts_speech"].shape[0] == 1
    vc = list(cosy.inference_vc(style, timbre, stream=False))
    assert len(vc) == 1
    # vc renders exactly the source's token count: duration follows the source, not the prompt
    src_tok = cosy.frontend.prompt(style).speech_tokens.shape[1]
    assert vc[0]["tts_speech"].shape[1] == cosy.cfg.mel_frames_for_tokens(src_tok) * cosy.cfg.upsample_total
    long = "Oh my god, it was just last weekend. " * 8
    assert len(list(cosy.inference_zero_shot(long, "I do.", style))) > 1     # split into segments, one yield each
    # stream=True: the same segment as consecutive chunks (2 s hops); a source longer than hop + overlap tokens gives several
    _tone(str(tmp_path / "long.wav"), 5.0, 180.0)
    src_long = load_wav(str(tmp_path / "long.wav"), 16000)
    n_tok = cosy.frontend.prompt(src_long).speech_tokens.shape[1]
    assert n_tok >= 240
    st = list(cosy.inference_vc(src_long, timbre, stream=True))
    assert len(st) == 1 + (n_tok - 120) // 100 + 1
    for o in st:
        c = o["tts_speech"]
        assert c.dtype == torch.float32 and c.device.type == "cpu" and c.shape[0] == 1 and c.shape[1] > 0
        assert torch.isfinite(c).all() and float(c.abs().max()) <= 0.99 + 1e-6
    total = sum(o["tts_speech"].shape[1] for o in st)
    assert abs(total / cosy.sample_rate - n_tok / cosy.cfg.token_rate) < 0.1
    short = list(cosy.inference_vc(style, timbre, stream=True))                 # 75 tokens: one (final) chunk
    assert len(short) == 1 and short[0]["tts_speech"].shape[1] == vc[0]["tts_speech"].shape[1]


def test_tts_with_rag_driver_end_to_end(cosy, tmp_path, golden_dir):
    from astts import audio
    from astts.cli import tts_with_rag as drv

    timbre_dir = tmp_path / "timbre"
    timbre_dir.mkdir()
    for sp, f in (("w1", 300.0), ("m2", 140.0)):
        _tone(str(timbre_dir / drv.TIMBRE_FILES[sp]), 1.0, f)
    _tone(str(timbre_dir / drv.WHISPER_TIMBRE_FILE), 1.0, 500.0)
    rows = [json.loads(l) for l in open(os.path.join(golden_dir, "search_results.jsonl"), encoding="utf-8")]
    picked = [r for r in rows if r["speaker"] in ("w1", "m2")][:2]
    picked.append(dict(picked[0], whisper=1))
    for i, r in enumerate(picked):                                  # retrieved style wavs live where the record says
        p = tmp_path / "seg" / (os.path.basename(r["retrieved_file_id"]) + ".wav")
        p.parent.mkdir(exist_ok=True)
        _tone(str(p), 1.0, 200.0 + 10 * i)
        r["retrieved_file_id"] = str(p)
    corr = tmp_path / "corr.jsonl"
    corr.write_text("\n".join(json.dumps(r, ensure_ascii=False) for r in picked) + "\n", encoding="utf-8")
    args = drv.build_parser().parse_args(["--corresponding_json", str(corr), "--result_dir", str(tmp_path / "res"),
                                          "--timbre_dir", str(timbre_dir), "--whisper_timbre_wav", str(timbre_dir / drv.WHISPER_TIMBRE_FILE)])
    now = datetime(2026, 3, 13, 14, 5)
    written = drv.tts_for_infer(args, cosyvoice=cosy, now=now)
    assert os.path.dirname(written[0]) == str(tmp_path / "res") + "_03131405"          # _%m%d%H%M suffix
    names = [os.path.basename(w) for w in written]
    for cnt, r in enumerate(picked, start=1):
        fid = os.path.basename(r["retrieved_file_id"])[:-4]
        assert f"{cnt}_{fid}_to_{r['speaker']}_0.wav" in names
    x, sr = audio.read_wav(written[0])
    assert sr == 22050 and x.shape[0] == 1 and np.isfinite(x).all() and np.abs(x).max() <= 0.99 + 1e-6
    # batched schedule (--batch_size): same file names; every row's duration stays inside its own 2x..20x window
    args_b = drv.build_parser().parse_args(["--corresponding_json", str(corr), "--result_dir", str(tmp_path / "resb"), "--batch_size", "3",
                                            "--timbre_dir", str(timbre_dir), "--whisper_timbre_wav", str(timbre_dir / drv.WHISPER_TIMBRE_FILE)])
    written_b = drv.tts_for_infer(args_b, cosyvoice=cosy, now=now)
    assert [os.path.basename(w) for w in written_b] == names
    for w, r in zip(written_b, picked):
        xb, srb = audio.read_wav(w)
        n_text = len(r["zh_text"].encode())
        frames = xb.shape[1] // cosy.cfg.upsample_total
        assert srb == 22050 and np.isfinite(xb).all() and np.abs(xb).max() <= 0.99 + 1e-6
        assert cosy.cfg.mel_frames_for_tokens(2 * n_text) <= frames <= cosy.cfg.mel_frames_for_tokens(20 * n_text)


def test_tts_with_style_and_timbre_driver(cosy, tmp_path):
    from astts import audio
    from astts.cli import tts_with_style_and_timbre as drv

    _tone(str(tmp_path / "pdd.wav"), 1.0, 250.0)
    _tone(str(tmp_path / "test80.wav"), 1.0, 180.0)
    (tmp_path / "lines.txt").write_text("Guess what?\nYes, I did it.\n", encoding="utf-8")
    args = drv.build_parser().parse_args(["--style_wav_path", str(tmp_path / "pdd.wav"), "--timbre_wav_path", str(tmp_path / "test80.wav"),
                                          "--style_wav_text", "When?", "--txt_path", str(tmp_path / "lines.txt"),
                                          "--result_dir", str(tmp_path / "out")])
    written = drv.tts_for_infer(args, cosyvoice=cosy)
    assert [os.path.basename(w) for w in written] == ["pdd_1_to_test80.wav", "pdd_2_to_test80.wav"]
    assert audio.read_wav(written[1])[1] == 22050
    exp = drv.tts_for_exp(args, cosyvoice=cosy)                      # zero-shot -> resample 22.05k->16k -> vc
    assert [os.path.basename(w) for w in exp] == ["pdd_1_to_test80_exp_0.wav", "pdd_2_to_test80_exp_0.wav"]


def test_search_embeddings_cli(golden_dir, tmp_path, capsys, kats):
    from astts.cli import search_embeddings as drv

    bank = np.load(os.path.join(golden_dir, "style_bank_130x6144.f16.npy")).astype(np.float32)
    q = tmp_path / "q.json"
    q.write_text(json.dumps(bank[29].tolist()))
    args = drv.build_parser().parse_args(["--query_embedding", str(q), "--db_path", os.path.join(golden_dir, "milvus_demo.db")])
    res = drv.main(args)
    out = capsys.readouterr().out
    assert [h["row"] for h in res[0]] == kats["self_top5_idx"][29][:3]
    assert "Top 3 results for Query 1:" in out and "File ID: " in out and "Distance: " in out
    # error convention of the reference wrapper: print + [] (search_embeddings.py:24-27)
    from astts.compat.pymilvus import MilvusClient
    assert drv.search_milvus(MilvusClient(os.path.join(golden_dir, "milvus_demo.db")), "missing", bank[0].tolist()) == []


def test_tts_for_dialog_driver(cosy, tmp_path):
    """tts_for_dialog.py:145-199: 1-based JSON-lines lookups, "null" entries skipped, one file per segment named
    {cnt}_{style_file_id}_to_{speaker}_{i}.wav under {result_dir}_{MMDDHHMM}."""
    from astts import audio
    from astts.cli import tts_for_dialog as drv

    os.makedirs(tmp_path / "styles")
    _tone(str(tmp_path / "styles" / "s_one.wav"), 1.2, 200.0)
    _tone(str(tmp_path / "styles" / "s_two.wav"), 1.0, 260.0)
    _tone(str(tmp_path / "spk_a.wav"), 1.1, 300.0)
    _tone(str(tmp_path / "spk_b.wav"), 1.1, 340.0)
    (tmp_path / "dialogue.jsonl").write_text("\n".join(json.dumps({"zh_text": t}) for t in ["Hello there.", "Never used.", "What now?"]) + "\n")
    (tmp_path / "styles.jsonl").write_text("\n".join(json.dumps({"file_id": f, "zh_text": t}) for f, t in
                                                     [("s_one", "First style."), ("s_two", "Second style.")]) + "\n")
    (tmp_path / "map.json").write_text(json.dumps({"1": {"value": 2, "speaker": "a", "emotion": "x"}, "2": "null",
                                                   "3": {"value": 1, "speaker": "b", "emotion": "y"}}))
    (tmp_path / "speakers.json").write_text(json.dumps({"a": str(tmp_path / "spk_a.wav"), "b": str(tmp_path / "spk_b.wav")}))
    argv = ["--corresponding_json", str(tmp_path / "map.json"), "--dialogue_json", str(tmp_path / "dialogue.jsonl"),
            "--style_wav_json", str(tmp_path / "styles.jsonl"), "--style_wav_dir", str(tmp_path / "styles"),
            "--result_dir", str(tmp_path / "out"), "--timbre_map", str(tmp_path / "speakers.json"), "--time_tag", "01020304"]
    written = drv.tts_for_infer(drv.build_parser().parse_args(argv), cosyvoice=cosy)
    names = sorted(os.path.basename(p) for p in written)
    assert names == ["1_s_two_to_a_0.wav", "2_s_one_to_b_0.wav"]
    assert all(os.path.dirname(p) == str(tmp_path / "out_01020304") for p in written)
    for p in written:
        w, sr = audio.read_wav(p)
        assert sr == 22050 and w.size > 0 and np.isfinite(w).all()
    exp = drv.tts_for_exp(drv.build_parser().parse_args(argv + ["--is_exp", "1"]), cosyvoice=cosy)
    assert sorted(os.path.basename(p) for p in exp) == ["s_one_0_to_b_exp_0.wav", "s_one_prompt_0_0.wav", "s_two_0_to_a_exp_0.wav",
                                                         "s_two_prompt_0_0.wav"]
    reader = drv.JsonDataReader(str(tmp_path / "styles.jsonl"))
    assert reader.get_zh_text_by_index(1) == "First style." and reader.get_fileid(2) == "s_two"


def test_vc_from_dir_driver(cosy, tmp_path):
    """vc_from_dir.py:79-220: style x timbre x line sweep, {style}_to_{timbre}_{cnt}_new.wav, meta.lst rows
    name|style_text|timbre_path|line; vc_from_dir_seed.py's meta.lst style source."""
    from astts.cli import vc_from_dir as drv

    os.makedirs(tmp_path / "styles")
    os.makedirs(tmp_path / "timbres")
    for n, f in (("aa", 210.0), ("bb", 250.0), ("cc", 290.0)):
        _tone(str(tmp_path / "styles" / f"{n}.wav"), 1.0, f)
    for n, f in (("t1", 310.0), ("t2", 350.0)):
        _tone(str(tmp_path / "timbres" / f"{n}.wav"), 1.0, f)
    (tmp_path / "styles.json").write_text(json.dumps([{"file_id": f"denoise_{n}", "zh_text": f"text of {n}"} for n in ("aa", "bb", "cc")]))
    (tmp_path / "lines.txt").write_text("One line.\nAnother line.\n")
    argv = ["--txt_path", str(tmp_path / "lines.txt"), "--style_dir", str(tmp_path / "styles"), "--timbre_dir", str(tmp_path / "timbres"),
            "--result_dir", str(tmp_path / "out"), "--style_num", "2", "--timbre_num", "2", "--style_json", str(tmp_path / "styles.json"),
            "--seed", "3"]
    rows = drv.main(argv, cosyvoice=cosy)
    assert len(rows) == 2 * 2 * 2
    meta = (tmp_path / "out" / "meta.lst").read_text().strip().split("\n")
    assert len(meta) == 8
    for row, line in zip(rows, meta):
        name, style_text, timbre_path, text = line.split("|")
        assert [name, style_text, timbre_path, text] == row
        style, rest = name.split("_to_")
        assert style in ("aa", "bb", "cc") and style_text == f"text of {style}" and rest.endswith("_new")
        assert os.path.exists(tmp_path / "out" / (name + ".wav")) and os.path.exists(timbre_path)
        assert text in ("One line.", "Another line.")
    with pytest.raises(ValueError):
        drv.get_path(str(tmp_path / "timbres"), 5)
    # seed-tts variant: styles and their transcripts from a meta.lst
    (tmp_path / "seed.lst").write_text(f"u1|seed text one|{tmp_path}/styles/aa.wav|x\nu2|seed text two|{tmp_path}/styles/bb.wav|y\n")
    rows2 = drv.main(["--txt_path", str(tmp_path / "lines.txt"), "--style_dir", str(tmp_path / "styles"), "--timbre_dir",
                      str(tmp_path / "timbres"), "--result_dir", str(tmp_path / "out2"), "--style_num", "1", "--timbre_num", "1",
                      "--style_meta_lst", str(tmp_path / "seed.lst"), "--seed", "1"], cosyvoice=cosy)
    assert len(rows2) == 2 and rows2[0][1] in ("seed text one", "seed text two")


def test_batch_surface_does_not_depend_on_its_schedule(cosy, tmp_path):
    """`synthesize_batch` runs its LM jobs on two worker threads (longest first) and renders groups as their tokens arrive.  A request's
    tokens must not depend on that schedule: the same requests with per-request seeds through (a) one 40-row call = two LM jobs, two
    render groups, (b) render groups of 8 rows = five LM jobs, (c) every request alone -- tokens equal, audio equal up to the rounding
    of other GEMM tiles (tiny model: the prefix comes out of other tiles in another batch)."""
    from astts.compat.cosyvoice import load_wav

    _tone(str(tmp_path / "s.wav"), 1.5, 220.0)
    _tone(str(tmp_path / "t.wav"), 1.2, 330.0)
    style, timbre = load_wav(str(tmp_path / "s.wav"), 16000), load_wav(str(tmp_path / "t.wav"), 16000)
    texts = [("line %d " % i) + "word " * (1 + (i * 7) % 11) for i in range(40)]
    items = [(t, "He did.", style, timbre) for t in texts]
    seeds = [1000 + i for i in range(40)]
    fixed = [20 + (i * 13) % 50 for i in range(40)]

    def run(max_batch, sel=None):
        idx = list(range(40)) if sel is None else sel
        outs = cosy.inference_tts_with_st_batch([items[i] for i in idx], max_batch=max_batch, split=False, seeds=[seeds[i] for i in idx],
                                                fixed_tokens=[fixed[i] for i in idx])
        return [t.clone() for t in cosy.last_tokens], [o[0]["tts_speech"] for o in outs]

    ta, wa = run(32)
    tb, wb = run(8)
    assert all(int(t.numel()) == f for t, f in zip(ta, fixed))
    agree = sum(int(torch.equal(x, y)) for x, y in zip(ta, tb))
    assert agree >= 36, agree                                # free-running sampling can amplify a 1-ulp prefix difference at a near-tie
    for i in (0, 17, 39):
        t1, w1 = run(1, [i])
        if torch.equal(t1[0], ta[i]):
            a, b = w1[0].double(), wa[i].double()
            snr = 10.0 * np.log10(float((b ** 2).sum()) / max(float(((a - b) ** 2).sum()), 1e-30))
            assert snr > 30.0, (i, snr)
    # the default (unseeded) path is deterministic given the instance generator's state, whatever the schedule
    g0 = cosy._gen.get_state()
    cosy.synthesize_batch  # noqa: B018  (the surface under test)
    o1 = cosy.inference_tts_with_st_batch(items[:36], max_batch=32, split=False, fixed_tokens=fixed[:36])
    t1 = [t.clone() for t in cosy.last_tokens]
    cosy._gen.set_state(g0)
    o2 = cosy.inference_tts_with_st_batch(items[:36], max_batch=32, split=False, fixed_tokens=fixed[:36])
    assert all(torch.equal(x, y) for x, y in zip(t1, cosy.last_tokens))
    assert all(torch.equal(x[0]["tts_speech"], y[0]["tts_speech"]) for x, y in zip(o1, o2))


def _stream_inputs():
    g = torch.Generator().manual_seed(3)
    t16 = torch.arange(int(1.5 * 16000)) / 16000
    style = (0.3 * torch.sin(2 * np.pi * 220 * t16) + 0.01 * torch.randn(t16.shape, generator=g))[None]
    timbre = (0.3 * torch.sin(2 * np.pi * 330 * t16) + 0.01 * torch.randn(t16.shape, generator=g))[None]
    return style, timbre


def test_stream_true_live_decode_yields_the_chunks_of_the_one_pass_form(cosy):
    """stream=True with the LM decoding hop by hop on its own stream while the chunks render (astts_lm_decode_range, _LmTokenStream)
    yields BIT-IDENTICAL chunks to decoding the whole segment first and chunking afterwards (rounds 3-4): same prefix, same uniforms,
    same sampler history across the ranges, same render draws -- fixed-length segments (3 chunks) and EOS-terminated ones."""
    import warnings

    style, timbre = _stream_inputs()
    args = ("I did it, I asked her to marry me.", "He did. In Niagara Falls.", style, timbre)
    for kw in ({"fixed_tokens": 230}, {}):
        outs = {}
        for live in (True, False):
            cosy.stream_lm_live = live
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)          # the tiny model's 512-position tables cap the 20x window
                outs[live] = [o["tts_speech"] for o in cosy.inference_tts_with_st(*args, stream=True, seed=11, **kw)]
        cosy.stream_lm_live = True
        assert len(outs[True]) == len(outs[False]) and len(outs[True]) >= 1
        if kw:
            assert len(outs[True]) == 3                                   # 230 tokens: hops at 0 and 100 + 30 left over
        for a, b in zip(outs[True], outs[False]):
            assert a.shape == b.shape and torch.equal(a, b)
    # ... and a generator abandoned after its first chunk shuts its decode worker down (no thread, no stream left waiting)
    it = cosy.inference_tts_with_st(*args, stream=True, seed=11, fixed_tokens=230)
    first = next(it)
    it.close()
    assert first["tts_speech"].shape[1] > 0
    again = [o["tts_speech"] for o in cosy.inference_tts_with_st(*args, stream=True, seed=11, fixed_tokens=230)]
    assert torch.equal(again[0], first["tts_speech"])


def test_stream_true_first_chunk_arrives_while_the_lm_still_decodes():
    """Time to the first yielded chunk at the CosyVoice-300M widths: upstream's schedule needs hop + look-ahead = 120 tokens before
    chunk 0, so 120 / n of the decode is the floor; the one-pass form (decode to the end, then chunk) pays the whole decode first.
    250-token segment: first chunk before 70 % of the segment's wall time (observed 0.55-0.57) AND earlier than the one-pass form's; 500
    tokens: before 45 % (observed 0.36)."""
    import time
    import warnings

    from astts.compat.cosyvoice import CosyVoice
    from astts.synth.config import SynthConfig

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cv = CosyVoice("/nonexistent", config=SynthConfig(), seed=0, allow_random_init=True)
    style, timbre = _stream_inputs()
    args = ("I did it, I asked her to marry me.", "He did. In Niagara Falls.", style, timbre)

    def run(live, n_tok):
        cv.stream_lm_live = live
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = None
        for _ in cv.inference_tts_with_st(*args, stream=True, seed=2, fixed_tokens=n_tok):
            first = first if first is not None else time.perf_counter() - t0
        return first, time.perf_counter() - t0

    for n_tok, bar in ((250, 0.70), (500, 0.45)):
        run(True, n_tok), run(False, n_tok)                                # warm-up
        live = min((run(True, n_tok) for _ in range(3)), key=lambda r: r[0])
        once = min((run(False, n_tok) for _ in range(3)), key=lambda r: r[0])
        print(f"[stream] {n_tok} tokens: first chunk {live[0] * 1e3:.1f} ms of {live[1] * 1e3:.1f} ms ({live[0] / live[1]:.2f}); "
              f"one-pass form {once[0] * 1e3:.1f} of {once[1] * 1e3:.1f} ms")
        assert live[0] / live[1] <= bar, (n_tok, live, once)
        assert live[0] < 0.9 * once[0], (n_tok, live, once)


def test_cosyvoice_from_a_checkpoint_directory_with_its_json_config(tmp_path):
    """CosyVoice(model_dir) as the reference calls it (tts_with_rag.py:159) on a directory that holds llm.pt / flow.pt / hift.pt in
    upstream's saved form (weight_norm pairs, extra buffers) + this build's astts.json: config from the file (24 kHz here), weights
    checked against its manifest and loaded, audio equal to the same weights handed over in memory."""
    from test_checkpoint_cpu import _write_model_dir

    from astts.compat.cosyvoice import CosyVoice
    from astts.synth.config import SynthConfig
    from astts.synth.model import SynthEngine
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny().with_(sample_rate=24000)
    state = make_all(cfg, 9)
    d = str(tmp_path / "CosyVoice-300M")
    _write_model_dir(d, cfg, state)
    # loaded synthesis weights without the frontend's files: refused (byte ids / untrained tokens into trained networks = garbage, status OK)
    with pytest.raises(FileNotFoundError) as ei:
        CosyVoice(d, seed=4)
    assert "speech_tokenizer_v1" in str(ei.value) and "campplus" in str(ei.value) and "tiktoken" in str(ei.value)
    cv = CosyVoice(d, seed=4, allow_standin_frontend=True)
    assert cv.random_init is False and cv.sample_rate == 24000 and cv.cfg == cfg
    assert cv.frontend.describe() == {"text_tokenizer": "stand-in", "speech_tokenizer": "synthetic-weights", "speaker_embedder": "synthetic-weights"}
    ref = CosyVoice("/nonexistent", config=cfg, seed=4, allow_random_init=True, engine=SynthEngine(state, cfg))
    style, timbre = _stream_inputs()
    a = list(cv.inference_tts_with_st("Guess what?", "I do. Yeah.", style, timbre, seed=3))
    b = list(ref.inference_tts_with_st("Guess what?", "I do. Yeah.", style, timbre, seed=3))
    assert len(a) == len(b) == 1 and a[0]["tts_speech"].shape == b[0]["tts_speech"].shape
    err = float((a[0]["tts_speech"] - b[0]["tts_speech"]).abs().max())
    print(f"[checkpoint dir] max waveform difference vs the in-memory weights {err:.2e}")
    assert err < 5e-2, err            # the folded weights differ from the originals by fp32 rounding of v * (g / |v|) (f0 -> phase amplifies it)


def test_batch_surface_with_wide_lm_jobs_spanning_render_groups(cosy, tmp_path):
    """Throughput form of the batch surface (`CosyVoice.wide_lm`, `lm_rows` > 32): LM jobs on the engine's wide path, decoupled from the
    render groups (a job spans several groups, a group several jobs).  With the tokens teacher-forced the LM's rounding is out of the
    picture, so every row must come back BIT-IDENTICAL to the default form (32-row jobs) -- same render groups, same draws: what is
    held here is the bookkeeping (which tokens reach which group, completion-order rendering, asynchronous result copies).  Free
    running, tokens stay valid and most rows agree with the 32-row path (near-ties of the sampler may differ: another summation order)."""
    from astts.compat.cosyvoice import load_wav

    _tone(str(tmp_path / "s.wav"), 1.5, 220.0)
    _tone(str(tmp_path / "t.wav"), 1.2, 330.0)
    style, timbre = load_wav(str(tmp_path / "s.wav"), 16000), load_wav(str(tmp_path / "t.wav"), 16000)
    n = 70
    items = [(("row %d " % i) + "word " * (1 + (i * 5) % 9), "He did.", style, timbre) for i in range(n)]
    seeds = [500 + i for i in range(n)]
    fixed = [12 + (i * 11) % 40 for i in range(n)]
    g = torch.Generator().manual_seed(8)
    forced = [torch.randint(0, cosy.cfg.speech_vocab, (f,), generator=g) for f in fixed]

    def run(wide, lm_rows, max_batch, forced_tokens):
        cosy.wide_lm, cosy.lm_rows = wide, lm_rows
        try:
            outs = cosy.inference_tts_with_st_batch(items, max_batch=max_batch, split=False, seeds=seeds, fixed_tokens=fixed, forced=forced_tokens)
        finally:
            cosy.wide_lm, cosy.lm_rows = False, 32
        return [t.clone() for t in cosy.last_tokens], [o[0]["tts_speech"] for o in outs]

    t0, w0 = run(False, 32, 16, forced)
    t1, w1 = run(True, 35, 16, forced)                     # two 35-row jobs over five 16-row render groups (14 rows in the last)
    assert all(torch.equal(a, f.to(torch.int32)) for a, f in zip(t1, forced))
    for i in range(n):
        assert w1[i].shape == w0[i].shape and torch.equal(w1[i], w0[i]), i
    tf0, _ = run(False, 32, 16, None)
    tf1, wf1 = run(True, 70, 16, None)                     # one 70-row job
    assert all(int(t.max()) < cosy.cfg.speech_vocab and int(t.numel()) == f for t, f in zip(tf1, fixed))
    agree = sum(int(torch.equal(a, b)) for a, b in zip(tf0, tf1))
    print(f"[wide lm] free-running rows whose tokens equal the 32-row path: {agree} of {n}")
    assert agree >= n // 2, agree
    assert all(bool(torch.isfinite(w).all()) for w in wf1)


def test_cosyvoice_wires_its_frontend_from_the_model_directory(tmp_path):
    """A model directory that holds everything the reference's holds (tts_with_rag.py:159 [EXT]: llm / flow / hift weights, the BPE
    vocabulary, the speech tokenizer and the speaker network -- the last two once as .pt state dicts, once as ONNX initializers):
    CosyVoice(model_dir) loads all of it without a flag, text ids come from the BPE, prompt features from the loaded networks (equal to
    the same networks built in memory), and the directory WITHOUT one of the files is refused."""
    import os

    from test_checkpoint_cpu import _write_model_dir

    from astts.bpe import TiktokenBPE

    from astts import frontend_nets as fn
    from astts import frontend_weights as fw
    from astts.compat.cosyvoice import CosyVoice
    from astts.onnx_weights import write_initializers
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig.tiny()
    d = str(tmp_path / "CosyVoice-300M")
    _write_model_dir(d, cfg, make_all(cfg, 9))
    tshape, cshape = fn.shapes_for(cfg)
    tsd, csd = fw.make_speech_tokenizer_weights(tshape, 41), fw.make_campplus_weights(cshape, 42)
    torch.save(tsd, os.path.join(d, "speech_tokenizer_v1.pt"))
    write_initializers(os.path.join(d, "campplus.onnx"), [(k, v.numpy()) for k, v in csd.items()])
    ranks = {bytes([b]): b for b in range(256)}                       # a byte-level vocabulary with a few merges (tiny config: 300 text ids)
    for i, m in enumerate([b"he", b"ll", b"llo", b"hello", b" w", b"or", b" wor", b"ld"]):
        ranks[m] = 256 + i
    bpe = TiktokenBPE(ranks)
    bpe.to_file(os.path.join(d, "multilingual.tiktoken"))
    cv = CosyVoice(d, seed=1)
    assert cv.frontend.describe() == {"text_tokenizer": "loaded", "speech_tokenizer": "loaded", "speaker_embedder": "loaded"}
    assert cv.frontend.text_ids("hello world")[0].tolist() == bpe.encode("hello world")
    style, _ = _stream_inputs()
    feats = cv.frontend.prompt(style)
    tok = fn.SpeechTokenizerV1(tsd, tshape, "cuda")(style)
    emb = fn.CamPlusSpeakerNet(csd, cshape, "cuda")(style)
    assert torch.equal(feats.speech_tokens, tok[:, :feats.speech_tokens.shape[1]]) and torch.equal(feats.spk_embedding, emb)
    assert int(feats.speech_tokens.max()) < cfg.speech_vocab and feats.spk_embedding.shape == (1, cfg.spk_dim)
    out = list(cv.inference_tts_with_st("Guess what?", "I do. Yeah.", style, style, seed=3, fixed_tokens=20))
    assert len(out) == 1 and bool(torch.isfinite(out[0]["tts_speech"]).all())
    # many prompts at once: equal lengths run as one GPU batch; every row is what prompt() gives it (up to GEMM tile order)
    other = torch.roll(style, 777, dims=1) * 0.8
    short = style[:, :20000].contiguous()
    many = cv.frontend.prompts([style, short, other, style])
    for w, f in zip([style, short, other, style], many):
        one = cv.frontend.prompt(w)
        assert f.speech_tokens.shape == one.speech_tokens.shape and f.mel.shape == one.mel.shape
        assert float((f.speech_tokens == one.speech_tokens).float().mean()) >= 0.97
        assert torch.allclose(f.spk_embedding, one.spk_embedding, atol=2e-3 * float(one.spk_embedding.abs().max()))
        assert torch.allclose(f.mel, one.mel, atol=1e-4)
    os.remove(os.path.join(d, "campplus.onnx"))
    with pytest.raises(FileNotFoundError) as ei:
        CosyVoice(d, seed=1)
    assert "campplus" in str(ei.value) and "speech_tokenizer_v1" not in str(ei.value)
    # a mis-shaped weight file is reported tensor by tensor
    bad = dict(csd)
    bad["xvector.dense.linear.weight"] = bad["xvector.dense.linear.weight"][:5]
    torch.save(bad, os.path.join(d, "campplus.pt"))
    with pytest.raises(ValueError) as ei:
        CosyVoice(d, seed=1)
    assert "xvector.dense.linear.weight" in str(ei.value)
