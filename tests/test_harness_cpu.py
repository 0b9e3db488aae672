"""CPU tests of the driver-script restatements and the host-side audio/front-end plumbing (SURVEY.md 8c
"Python-harness rows": a7, a9, a12, a16, a17), against the committed copy of the reference's own hand-off file."""
import json
import os

import numpy as np
import pytest
import torch


def test_work_items_from_recorded_handoff(golden_dir):
    from astts.cli import tts_with_rag as drv

    items = drv.get_text_and_wav(os.path.join(golden_dir, "search_results.jsonl"))
    rows = [json.loads(l) for l in open(os.path.join(golden_dir, "search_results.jsonl"), encoding="utf-8")]
    assert len(items) == len(rows) == 64
    for it, r in zip(items, rows):
        assert it["tts_text"] == r["zh_text"] and it["speaker"] == r["speaker"]
        assert it["style_wav_path"] == r["retrieved_file_id"] and it["style_wav_text"] == r["retrieved_text"]
        assert it["is_whisper"] == r["whisper"]
        assert it["timbre_wav_path"].endswith(f"_{r['speaker']}.wav")            # w1/w2/m1/m2 mapping, tts_with_rag.py:66-75
        assert it["timbre_wav_path"].startswith(drv.REF_TIMBRE_DIR)
    assert {it["speaker"] for it in items} <= {"w1", "w2", "m1", "m2"}
    assert set(items[0]) == {"is_whisper", "tts_text", "speaker", "timbre_wav_path", "style_wav_path", "style_wav_text"}
    # output naming: f'{cnt}_{style_wav_fileid}_to_{timbre}' + '_{}.wav'.format(i); [:-4] strips 4 chars whatever they are
    assert drv.output_name(3, "/x/seg_wav/tonight1/tonight1_0h5m5dot0s_0h5m6dot51s.wav", "w1", 0) == \
        "3_tonight1_0h5m5dot0s_0h5m6dot51s_to_w1_0.wav"
    assert drv.output_name(1, rows[0]["retrieved_file_id"], "w1", 2) == \
        f"1_{os.path.basename(rows[0]['retrieved_file_id'])[:-4]}_to_w1_2.wav"
    with pytest.raises(KeyError):
        drv.get_timbre_wav_path("jinjing")


def test_cli_flags_match_reference():
    from astts.cli import search_embeddings, tts_with_rag, tts_with_style_and_timbre

    a = tts_with_rag.build_parser().parse_args(["--corresponding_json", "x.json", "--result_dir", "out"])
    assert a.is_exp is False and a.corresponding_json == "x.json"
    assert tts_with_rag.build_parser().parse_args(["--corresponding_json", "x", "--result_dir", "o", "--is_exp", "False"]).is_exp is True  # type=bool quirk
    b = tts_with_style_and_timbre.build_parser().parse_args(["--style_wav_path", "s.wav", "--timbre_wav_path", "t.wav",
                                                             "--style_wav_text", "hi", "--txt_path", "a.txt", "--result_dir", "o"])
    assert b.style_wav_text == "hi" and b.is_exp is False
    c = search_embeddings.build_parser().parse_args(["--query_embedding", "q.json"])
    assert c.top_k == 3 and c.db_path == "milvus_demo.db"


def test_wav_roundtrip_and_formats(tmp_path):
    from astts import audio

    x = torch.sin(torch.linspace(0, 200, 22050))[None, :] * 0.5
    p = str(tmp_path / "a.wav")
    audio.write_wav(p, x, 22050)
    y, sr = audio.read_wav(p)
    assert sr == 22050 and np.array_equal(y, x.numpy())                      # float32 WAVE is lossless
    # PCM16 stereo written by the stdlib, read back as mono mean by load_wav
    import wave
    p2 = str(tmp_path / "b.wav")
    pcm = (np.stack([x[0].numpy(), -0.5 * x[0].numpy()], axis=1) * 32767).astype("<i2")
    with wave.open(p2, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    m = audio.load_wav(p2, 22050)
    assert m.shape == (1, 22050) and float((m[0] - 0.25 * x[0]).abs().max()) < 1e-4
    r = audio.load_wav(p2, 16000)
    assert r.shape == (1, 16000)


def test_resampler_preserves_in_band_tone():
    from astts import audio

    sr_in, sr_out, f = 22050, 16000, 1000.0
    t = torch.arange(sr_in) / sr_in
    y = audio.resample(torch.sin(2 * np.pi * f * t)[None, :], sr_in, sr_out)
    assert y.shape == (1, sr_out)
    t2 = torch.arange(sr_out) / sr_out
    err = (y[0, 200:-200] - torch.sin(2 * np.pi * f * t2)[200:-200]).abs().max()
    assert float(err) < 2e-2
    assert audio.resample(y, 16000, 16000) is y


def test_mel_and_frontend_shapes():
    from astts import audio
    from astts.frontend import ByteTokenizer, Frontend, text_normalize
    from astts.synth.config import SynthConfig

    fb = audio.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert fb.shape == (80, 513) and fb.min() >= 0 and np.all(fb.sum(axis=1) > 0)
    assert np.all(fb[:, int(8000 / (22050 / 2) * 512) + 2:] == 0)              # nothing above fmax
    cfg = SynthConfig.tiny()
    fe = Frontend(cfg)
    wav = torch.randn(1, 3 * 16000) * 0.1
    pf = fe.prompt(wav)
    assert pf.speech_tokens.dtype == torch.int32 and abs(pf.speech_tokens.shape[1] - 150) <= 2   # 50 Hz
    assert pf.spk_embedding.shape == (1, cfg.spk_dim)
    assert pf.mel.shape[2] == 80 and pf.mel.shape[1] == cfg.mel_frames_for_tokens(pf.speech_tokens.shape[1])
    assert int(pf.speech_tokens.max()) < cfg.speech_vocab
    pf2 = fe.prompt(wav)
    assert torch.equal(pf.speech_tokens, pf2.speech_tokens)                    # deterministic stand-ins
    tok = ByteTokenizer(cfg.text_vocab)
    segs = text_normalize("Guess what? I did it, I asked her to marry me. " * 12, tok, split=True)
    assert len(segs) > 1 and all(len(tok.encode(s)) <= 80 + 60 for s in segs)
    assert text_normalize("what?", tok) == ["what?"]
    assert fe.text_ids("I did it").shape == (1, 8)


def test_search_json_record_format(golden_dir, tmp_path, monkeypatch):
    """The JSONL the retrieval driver writes is exactly what get_text_and_wav consumes (+ the hand-added whisper key)."""
    from astts.cli import search_json
    from astts.compat import pymilvus as pm

    bank = np.load(os.path.join(golden_dir, "style_bank_130x6144.f16.npy")).astype(np.float32)
    meta = json.load(open(os.path.join(golden_dir, "style_bank_meta.json")))

    class OracleBank:                                                          # CPU stand-in for the GPU bank (checker only)
        def __init__(self, m):
            self.m = m

        def search(self, q, k):
            from oracle import knn as oknn
            idx, sc = oknn.knn_search(self.m.astype(np.float16), np.asarray(q, np.float32), k)
            return idx, sc.astype(np.float32)

    monkeypatch.setattr(pm._Collection, "bank", lambda self: OracleBank(self.matrix()))
    inp = tmp_path / "in.jsonl"
    inp.write_text("\n".join(json.dumps({"zh_text": f"line {i}", "speaker": "w1"}) for i in range(3)) + "\n", encoding="utf-8")
    np.save(tmp_path / "q.npy", bank[[5, 61, 129]])
    out = tmp_path / "out" / "search_results.json"
    args = search_json.build_parser().parse_args(["--input_json", str(inp), "--query_npy", str(tmp_path / "q.npy"),
                                                  "--db_path", os.path.join(golden_dir, "milvus_demo.db"),
                                                  "--output_file", str(out), "--file_prefix_path", "/data/seg_wav"])
    res = search_json.main(args)
    rows = [json.loads(l) for l in open(out, encoding="utf-8")]
    assert rows == res and len(rows) == 3
    assert set(rows[0]) == {"zh_text", "speaker", "retrieved_file_id", "retrieved_text", "distance"}
    assert rows[1]["retrieved_file_id"] == "/data/seg_wav/" + meta["rows"][61]["file_id"]
    assert abs(rows[1]["distance"] - 1.0) < 1e-6


def test_search_json_skips_rows_without_text(golden_dir, tmp_path, monkeypatch):
    """milvus/search_json.py:385-387: a row whose zh_text is empty is skipped (no record), together with its query vector."""
    from astts.cli import search_json
    from astts.compat import pymilvus as pm

    bank = np.load(os.path.join(golden_dir, "style_bank_130x6144.f16.npy")).astype(np.float32)
    meta = json.load(open(os.path.join(golden_dir, "style_bank_meta.json")))
    seen = {}

    class OracleBank:
        def __init__(self, m):
            self.m = m

        def search(self, q, k):
            from oracle import knn as oknn
            seen["n"] = len(q)
            idx, sc = oknn.knn_search(self.m.astype(np.float16), np.asarray(q, np.float32), k)
            return idx, sc.astype(np.float32)

    monkeypatch.setattr(pm._Collection, "bank", lambda self: OracleBank(self.matrix()))
    inp = tmp_path / "in.jsonl"
    lines = [{"zh_text": "first", "speaker": "w1"}, {"zh_text": "   ", "speaker": "m1"}, {"speaker": "m2"}, {"zh_text": "last", "speaker": "w2"}]
    inp.write_text("\n".join(json.dumps(l) for l in lines) + "\n", encoding="utf-8")
    np.save(tmp_path / "q.npy", bank[[5, 6, 7, 129]])
    args = search_json.build_parser().parse_args(["--input_json", str(inp), "--query_npy", str(tmp_path / "q.npy"),
                                                  "--db_path", os.path.join(golden_dir, "milvus_demo.db")])
    res = search_json.main(args)
    assert seen["n"] == 2 and [r["zh_text"] for r in res] == ["first", "last"]
    assert res[1]["retrieved_file_id"] == meta["rows"][129]["file_id"]          # row 3's vector stayed with row 3


def test_search_json_llm_half_keeps_the_references_per_row_failure_semantics(golden_dir, tmp_path, monkeypatch, capsys):
    """milvus/search_json.py:389-404: a label that cannot be generated becomes "neutral", a row whose embedding fails gets an "Error"
    record -- and the rows around it still get their hits (one bad row failed the rank's whole shard)."""
    from astts.cli import search_json
    from astts.compat import pymilvus as pm

    bank = np.load(os.path.join(golden_dir, "style_bank_130x6144.f16.npy")).astype(np.float32)

    class OracleBank:
        def __init__(self, m):
            self.m = m

        def search(self, q, k):
            from oracle import knn as oknn
            idx, sc = oknn.knn_search(self.m.astype(np.float16), np.asarray(q, np.float32), k)
            return idx, sc.astype(np.float32)

    class Cfg:
        hidden = 3072

    class FlakyEmbedder:
        cfg = Cfg()

        def generate_emotion_labels(self, texts, max_new_tokens):
            if any("boom" in t for t in texts):
                raise RuntimeError("generation failed")
            return ["happy" if "!" in t else "sad" for t in texts]

        def get_embeddings(self, texts):
            if any(t == "sad" for t in texts):
                raise RuntimeError("embedding failed")
            return [bank[5, :3072] if t in ("happy", "neutral") else bank[5, 3072:] for t in texts]

    monkeypatch.setattr(pm._Collection, "bank", lambda self: OracleBank(self.matrix()))
    rows = [{"zh_text": "great!", "speaker": "w1"}, {"zh_text": "boom", "speaker": "w1"}, {"zh_text": "so so", "speaker": "w1"}]
    q, labels, failed = search_json.embed_rows(rows, FlakyEmbedder(), {}, batch=1)
    assert labels == ["happy", "neutral", "sad"] and failed.tolist() == [False, False, True]
    inp = tmp_path / "in.jsonl"
    inp.write_text("\n".join(json.dumps(r) for r in rows) + "\n", encoding="utf-8")
    args = search_json.build_parser().parse_args(["--input_json", str(inp), "--db_path", os.path.join(golden_dir, "milvus_demo.db"), "--llm_batch", "1"])
    res = search_json.main(args, embedder=FlakyEmbedder())
    assert [r["retrieved_file_id"] == "Error" for r in res] == [False, False, True]
    assert res[0]["retrieved_file_id"] == res[1]["retrieved_file_id"] != "N/A" and res[2]["distance"] == "Error"
    assert "Error during emotion generation" in capsys.readouterr().out


def test_cosyvoice_needs_weights_or_an_explicit_opt_in(tmp_path, monkeypatch):
    """A missing model_dir must not silently synthesise noise: the constructor raises before anything touches the
    GPU unless random-init weights are explicitly allowed."""
    from astts.compat.cosyvoice import CosyVoice

    monkeypatch.delenv("ASTTS_ALLOW_RANDOM_INIT", raising=False)
    with pytest.raises(FileNotFoundError, match="allow_random_init"):
        CosyVoice(str(tmp_path / "CosyVoice-300M"))
    for cli in ("tts_with_rag", "tts_with_style_and_timbre", "tts_for_dialog", "vc_from_dir"):
        mod = __import__(f"astts.cli.{cli}", fromlist=["x"])
        src = open(mod.__file__).read()
        assert "--allow_random_init" in src


def test_weight_norm_folding_both_checkpoint_forms():
    """a8: real CosyVoice checkpoints store weight-normalised convolutions as (weight_g, weight_v) -- torch.nn.utils.weight_norm
    -- or as parametrizations.weight.original0/1 -- torch.nn.utils.parametrizations.weight_norm.  Both fold to the module's
    effective weight, for Conv1d and ConvTranspose1d (norm over dim 0's complement, torch's default)."""
    import torch

    from astts.synth.weights import _fold_weight_norm

    torch.manual_seed(0)
    for make in (lambda: torch.nn.Conv1d(6, 10, 3), lambda: torch.nn.ConvTranspose1d(6, 10, 4, 2)):
        m_old = torch.nn.utils.weight_norm(make())
        with torch.no_grad():
            m_old.weight_g.mul_(1.7)
        x = torch.randn(2, 6, 9)
        y = m_old(x)
        sd = {f"conv.{k}": v for k, v in m_old.state_dict().items()}
        assert "conv.weight_g" in sd and "conv.weight_v" in sd
        folded = _fold_weight_norm(sd)
        assert set(folded) == {"conv.weight", "conv.bias"}
        m_plain = make()
        m_plain.load_state_dict({"weight": folded["conv.weight"], "bias": folded["conv.bias"]})
        assert torch.allclose(m_plain(x), y, atol=1e-6)
        m_new = torch.nn.utils.parametrizations.weight_norm(make())
        with torch.no_grad():
            m_new.parametrizations.weight.original0.mul_(0.6)
        y2 = m_new(x)
        sd2 = {f"conv.{k}": v for k, v in m_new.state_dict().items()}
        assert "conv.parametrizations.weight.original0" in sd2
        folded2 = _fold_weight_norm(sd2)
        assert set(folded2) == {"conv.weight", "conv.bias"}
        m_plain.load_state_dict({"weight": folded2["conv.weight"], "bias": folded2["conv.bias"]})
        assert torch.allclose(m_plain(x), y2, atol=1e-6)


def test_load_state_dicts_round_trip(tmp_path):
    """a8: make_all() -> llm.pt / flow.pt / hift.pt with the vocoder's convolutions re-expressed in both weight-norm forms ->
    load_state_dicts() returns the original tensors (the engine is then built from exactly the same state)."""
    import torch

    from astts.synth.config import SynthConfig
    from astts.synth.weights import load_state_dicts, make_all

    cfg = SynthConfig.tiny()
    state = make_all(cfg, 3)
    hift = dict(state["hift"])
    n_g, n_p = 0, 0
    for k in [k for k in hift if k.endswith(".weight") and hift[k].dim() == 3]:
        w = hift.pop(k)
        base = k[: -len(".weight")]
        norm = w.flatten(1).norm(dim=1).view(-1, 1, 1)
        if n_g <= n_p:
            hift[base + ".weight_g"], hift[base + ".weight_v"] = norm * 1.0, w.clone()
            n_g += 1
        else:
            hift[base + ".parametrizations.weight.original0"], hift[base + ".parametrizations.weight.original1"] = norm * 1.0, w / 2.0   # effective weight g v / norm(v) = w
            n_p += 1
    assert n_g >= 3 and n_p >= 3
    d = tmp_path / "CosyVoice-300M"
    d.mkdir()
    torch.save(state["llm"], d / "llm.pt")
    torch.save(state["flow"], d / "flow.pt")
    torch.save(hift, d / "hift.pt")
    got = load_state_dicts(str(d))
    for name in ("llm", "flow", "hift"):
        assert set(got[name]) == set(state[name]), name
        for k, v in state[name].items():
            assert torch.allclose(got[name][k], v.float(), atol=1e-6, rtol=1e-6), (name, k)
    (d / "flow.pt").unlink()
    with pytest.raises(FileNotFoundError):
        load_state_dicts(str(d))
