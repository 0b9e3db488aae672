"""The data-parallel form of the DRIVERS (BASELINE config 4: the IEMOCAP test set over the 8 GPUs of a node) on CPU: world-size 2 and 8
gloo process groups run astts.cli.search_json / astts.cli.tts_with_rag exactly as `python -m torch.distributed.run` would start them
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), with the GPU engines replaced by stand-ins:
  * retrieval: the real MilvusClient over the shipped bank, its StyleBank replaced by oracle/knn.py (checker use only);
  * synthesis: a stub CosyVoice whose "audio" is a pure function of (text, per-row seed).
Held: every rank works on its own row shard only (1 623 queries: seven shards of 203 and one of 202), the JSONL / the wav files are
identical to the one-process run's, file names carry the GLOBAL row number, all ranks write into ONE time-stamped directory.
Reference loops: /root/reference/milvus/search_json.py:382-461, /root/reference/tts_with_rag.py:172-197."""
import hashlib
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "autostyle-tts_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "ASTTS_DIST_BACKEND": "gloo"})
    torch.set_num_threads(1)


class _OracleBank:
    def __init__(self, m, log):
        self.m, self.log = m, log

    def search(self, q, k):
        from oracle import knn as oknn

        with open(self.log, "a") as f:
            f.write(f"{q.shape[0]}\n")
        idx, sc = oknn.knn_search(self.m.astype(np.float16), np.asarray(q, np.float32), k)
        return idx, sc.astype(np.float32)

    def close(self):
        pass


def _search_worker(rank, world, port, work):
    _env(rank, world, port)
    from astts import parallel
    from astts.cli import search_json
    from astts.compat import pymilvus as pm

    log = os.path.join(work, f"calls_w{world}_r{rank}.txt")
    pm._Collection.bank = lambda self: _OracleBank(self.matrix(), log)
    args = search_json.build_parser().parse_args(["--input_json", os.path.join(work, "in.jsonl"), "--query_npy", os.path.join(work, "q.npy"),
                                                  "--db_path", os.path.join(GOLD, "milvus_demo.db"),
                                                  "--output_file", os.path.join(work, f"out_w{world}.jsonl"), "--file_prefix_path", "/data/seg"])
    res = search_json.main(args)
    assert len(res) == 1623
    parallel.shutdown()


def _make_search_inputs(work):
    with open(os.path.join(GOLD, "iemocap_test_sentences.json")) as f:
        sents = json.load(f)
    sents = sents["all"]                     # the 1 623 sentences of data/iemocap.test.json, in file order
    assert len(sents) == 1623
    bank = np.load(os.path.join(GOLD, "style_bank_130x6144.f16.npy")).astype(np.float32)
    rng = np.random.default_rng(5)
    q = bank[rng.integers(0, 130, 1623)] + 0.5 * rng.standard_normal((1623, 6144)).astype(np.float32)
    np.save(os.path.join(work, "q.npy"), q.astype(np.float32))
    with open(os.path.join(work, "in.jsonl"), "w", encoding="utf-8") as f:
        for i, s in enumerate(sents):
            f.write(json.dumps({"zh_text": s if s.strip() else "x", "speaker": ["w1", "w2", "m1", "m2"][i % 4]}, ensure_ascii=False) + "\n")


@pytest.fixture(scope="module")
def search_work(tmp_path_factory):
    work = str(tmp_path_factory.mktemp("dist_search"))
    _make_search_inputs(work)
    mp.spawn(_search_worker, args=(1, _free_port(), work), nprocs=1, join=True)          # the one-process run (no process group)
    return work


@pytest.mark.parametrize("world", [2, 8])
def test_search_json_sharded_over_ranks_writes_the_one_process_file(search_work, world):
    work = search_work
    mp.spawn(_search_worker, args=(world, _free_port(), work), nprocs=world, join=True)
    one = open(os.path.join(work, "out_w1.jsonl"), "rb").read()
    many = open(os.path.join(work, f"out_w{world}.jsonl"), "rb").read()
    assert one == many and one.count(b"\n") == 1623
    per = (1623 + world - 1) // world
    sizes = [int(open(os.path.join(work, f"calls_w{world}_r{r}.txt")).read().split()[0]) for r in range(world)]
    assert sizes == [per] * (world - 1) + [1623 - per * (world - 1)]                      # world 8: 7 x 203 + 202
    assert [len(open(os.path.join(work, f"calls_w{world}_r{r}.txt")).read().split()) for r in range(world)] == [1] * world
    rec = json.loads(one.split(b"\n")[0])
    assert set(rec) == {"zh_text", "speaker", "retrieved_file_id", "retrieved_text", "distance"} and rec["retrieved_file_id"].startswith("/data/seg/")


class _StubEmbedder:
    """CPU stand-in for astts.llm.embedder.LlamaEmbedder (the GPU one is held to transformers in tests/test_llm_gpu.py): labels and
    3072-d vectors are pure functions of the text, and every call is logged so that the test can see WHICH rows a rank worked on."""

    class cfg:
        hidden = 3072

    def __init__(self, log):
        self.log = log

    @staticmethod
    def _vec(text):
        h = int.from_bytes(hashlib.sha256(text.encode("utf-8")).digest()[:8], "little")
        return np.random.default_rng(h).standard_normal(3072).astype(np.float32)

    def generate_emotion_labels(self, texts, max_new_tokens=10):
        with open(self.log, "a") as f:
            f.write(f"label {len(texts)}\n")
        return [["happy", "sad", "neutral", "angry", "excited", "frustrated"][len(t) % 6] for t in texts]

    def get_embeddings(self, texts):
        with open(self.log, "a") as f:
            f.write(f"embed {len(texts)}\n")
        return np.stack([self._vec(t) for t in texts])


def _search_llm_worker(rank, world, port, work):
    _env(rank, world, port)
    from astts import parallel
    from astts.cli import search_json
    from astts.compat import pymilvus as pm

    pm._Collection.bank = lambda self: _OracleBank(self.matrix(), os.path.join(work, f"llm_calls_w{world}_r{rank}.txt"))
    args = search_json.build_parser().parse_args(["--input_json", os.path.join(work, "in200.jsonl"), "--biography_json", os.path.join(work, "bios.json"),
                                                  "--db_path", os.path.join(GOLD, "milvus_demo.db"), "--llm_batch", "16",
                                                  "--output_file", os.path.join(work, f"llm_out_w{world}.jsonl")])
    res = search_json.main(args, embedder=_StubEmbedder(os.path.join(work, f"llm_embedder_w{world}_r{rank}.txt")))
    assert len(res) == 200
    parallel.shutdown()


@pytest.mark.parametrize("world", [2])
def test_search_json_llm_half_is_sharded_too(search_work, world):
    """milvus/search_json.py:372-411 (label -> embed -> search per row) under the data-parallel launch: a rank labels and embeds ITS rows
    only (no query vector crosses ranks: the all-gather carries ids and similarities), speakers without a biography get the reference's
    placeholder, and the JSONL equals the one-process run's."""
    work = search_work
    rows = [json.loads(l) for l in open(os.path.join(work, "in.jsonl"), encoding="utf-8")][:200]
    with open(os.path.join(work, "in200.jsonl"), "w", encoding="utf-8") as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")
    with open(os.path.join(work, "bios.json"), "w") as f:
        json.dump({"w1": "Cheerful and optimistic.", "m1": "Pragmatic and thoughtful."}, f)       # w2 / m2: the placeholder biography
    mp.spawn(_search_llm_worker, args=(1, _free_port(), work), nprocs=1, join=True)
    mp.spawn(_search_llm_worker, args=(world, _free_port(), work), nprocs=world, join=True)
    one = open(os.path.join(work, "llm_out_w1.jsonl"), "rb").read()
    many = open(os.path.join(work, f"llm_out_w{world}.jsonl"), "rb").read()
    assert one == many and one.count(b"\n") == 200
    for r in range(world):
        calls = open(os.path.join(work, f"llm_embedder_w{world}_r{r}.txt")).read().split("\n")
        labelled = sum(int(c.split()[1]) for c in calls if c.startswith("label"))
        assert labelled == 100                                                                    # its own shard, in batches of 16
        assert all(int(c.split()[1]) <= 16 for c in calls if c)
        embedded = sum(int(c.split()[1]) for c in calls if c.startswith("embed"))
        assert embedded <= 6 + 3                                                                  # distinct labels + distinct biographies, once each


class _StubVoice:
    """Audio = a pure function of (text, style text, per-row seed, segment): what the drivers may rely on across ranks."""
    sample_rate = 22050

    def __init__(self, log):
        self.log = log

    @staticmethod
    def _wav(text, style_text, seed, seg):
        h = hashlib.sha256(f"{text}|{style_text}|{seed}|{seg}".encode()).digest()
        g = torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))
        return (torch.rand(1, 400 + h[8], generator=g) * 2 - 1) * 0.5

    def inference_tts_with_st(self, tts_text, style_text, style_wav, timbre_wav, stream=False, seed=None):
        with open(self.log, "a") as f:
            f.write(tts_text + "\n")
        assert seed is not None
        for seg in range(1 + len(tts_text) % 2):
            yield {"tts_speech": self._wav(tts_text, style_text, seed, seg)}

    def inference_tts_with_st_batch(self, items, max_batch=32, seeds=None):
        assert seeds is not None and len(seeds) == len(items) and max_batch >= 1      # the surface groups the rows itself
        with open(self.log, "a") as f:
            for it in items:
                f.write(it[0] + "\n")
        return [[{"tts_speech": self._wav(it[0], it[1], s, seg)} for seg in range(1 + len(it[0]) % 2)] for it, s in zip(items, seeds)]


def _tts_worker(rank, world, port, work, bs):
    _env(rank, world, port)
    from datetime import datetime

    from astts import parallel
    from astts.cli import tts_with_rag
    from astts.compat import cosyvoice as cv

    cv.load_wav = lambda path, sr: torch.zeros(1, 160)                   # the style / timbre wavs of the recorded hand-off do not exist here
    args = tts_with_rag.build_parser().parse_args(["--corresponding_json", os.path.join(GOLD, "search_results.jsonl"),
                                                   "--result_dir", os.path.join(work, f"res_w{world}_b{bs}"), "--batch_size", str(bs), "--seed", "7"])
    # rank 0's time stamp names the directory for everyone: give the others a different clock
    now = datetime(2025, 3, 4, 5, 6) if rank == 0 else datetime(2025, 3, 4, 5, 7 + rank)
    written = tts_with_rag.tts_for_infer(args, cosyvoice=_StubVoice(os.path.join(work, f"rows_w{world}_b{bs}_r{rank}.txt")), now=now)
    assert all(os.path.dirname(p).endswith("_03040506") for p in written)
    parallel.shutdown()


@pytest.mark.parametrize("world,bs", [(2, 1), (8, 4)])
def test_tts_with_rag_sharded_over_ranks_writes_the_one_process_files(tmp_path, world, bs):
    work = str(tmp_path)
    mp.spawn(_tts_worker, args=(1, _free_port(), work, bs), nprocs=1, join=True)
    mp.spawn(_tts_worker, args=(world, _free_port(), work, bs), nprocs=world, join=True)
    d1, dw = os.path.join(work, f"res_w1_b{bs}_03040506"), os.path.join(work, f"res_w{world}_b{bs}_03040506")
    f1, fw = sorted(os.listdir(d1)), sorted(os.listdir(dw))
    assert f1 == fw and len(f1) >= 64
    assert {int(n.split("_")[0]) for n in f1} == set(range(1, 65))      # global row numbers 1..64 (tts_with_rag.py:172,196)
    for n in f1:
        assert open(os.path.join(d1, n), "rb").read() == open(os.path.join(dw, n), "rb").read(), n
    with open(os.path.join(GOLD, "search_results.jsonl"), encoding="utf-8") as f:
        texts = [json.loads(line)["zh_text"] for line in f if line.strip()]
    per = (64 + world - 1) // world
    for r in range(world):                                                # each rank saw exactly its shard, in order
        seen = open(os.path.join(work, f"rows_w{world}_b{bs}_r{r}.txt"), encoding="utf-8").read().split("\n")[:-1]
        assert seen == texts[r * per:(r + 1) * per]
