"""The path's collectives through librccl on the hardware that exists (ONE MI355X): `bench.py --force-dist` initialises
torch.distributed with backend "nccl" (= RCCL on ROCm) for a one-rank world, so the per-step all-gather of the retrieved
style ids (astts.parallel.gather_style_ids on the pipeline's front stream: north_star's only collective, replacing the
sequential loop of /root/reference/milvus/search_json.py:382-411) and the bank-sharded merge run through RCCL before an
8-GPU node ever sees them.  Fresh child process: the communicator must not share a process with the other GPU tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_one_rank_through_rccl():
    env = dict(os.environ, ASTTS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ASTTS_BENCH_STUB"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-24khz", "--no-side"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # stdout is the ONE JSON line and nothing else: RCCL prints a version banner to C stdout (flushed at exit, i.e. after the
    # line) -- bench.py hands file descriptor 1 to stderr for everything but its own line (json_only_stdout)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["rccl_world_size"] == 1 and res["rccl_backend"] == "nccl"
    assert res["ids_match_oracle"] is True
    assert res["gathered_ids_match_oracle"] is True          # ids after all_gather_into_tensor == the oracle's
    assert res["bank_sharded_ids_match_oracle"] is True      # (row, fp64 score) all-gather + local merge == unsharded ids
    assert res["waveform_finite_and_clamped"] is True
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["value"] > 0
    print("rccl one-rank bench:", {k: res[k] for k in ("value", "ms_per_step", "rccl_world_size", "rccl_backend")})


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_prints_one_json_line():
    """The driver's multi-GPU form (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N), with the N this box
    has: ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher, rank 0's stdout carries exactly one line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ASTTS_BENCH_FORCE_DIST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "ASTTS_BENCH_STUB"):
        env.pop(k, None)
    sys.path.insert(0, ROOT)
    from bench import free_port

    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-24khz", "--no-cobatch", "--no-side"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["rccl_world_size"] == 1 and res["rccl_backend"] == "nccl" and res["gathered_ids_match_oracle"] is True
    assert res["n_gpus"] == 1 and res["value"] > 0


@pytest.mark.gpu
def test_drivers_under_torch_distributed_run_through_rccl(tmp_path):
    """The retrieval DRIVER's data-parallel form on the hardware that exists: `python -m torch.distributed.run --nproc-per-node 1 -m
    astts.cli.search_json` with ASTTS_FORCE_DIST=1 forms a one-rank RCCL process group (backend nccl), so the all-gather of the (style
    id, similarity) pairs and the barrier run through librccl on the GPU; the JSONL equals the plain one-process run's, byte for byte.
    (World sizes 2 and 8, and the synthesis driver: tests/test_drivers_dist_cpu.py over gloo.)"""
    import numpy as np

    sys.path.insert(0, ROOT)
    from bench import free_port

    gold = os.path.join(ROOT, "tests", "golden")
    bank = np.load(os.path.join(gold, "style_bank_130x6144.f16.npy")).astype(np.float32)
    rng = np.random.default_rng(3)
    rows = rng.integers(0, 130, 37)
    np.save(tmp_path / "q.npy", (bank[rows] + 0.3 * rng.standard_normal((37, 6144))).astype(np.float32))
    (tmp_path / "in.jsonl").write_text("\n".join(json.dumps({"zh_text": f"line {i}", "speaker": ["w1", "m2"][i % 2]}) for i in range(37)) + "\n")
    pkg = os.path.join(ROOT, "autostyle-tts_amd")
    base_env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=pkg + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        base_env.pop(k, None)
    args = ["--input_json", str(tmp_path / "in.jsonl"), "--query_npy", str(tmp_path / "q.npy"), "--db_path", os.path.join(gold, "milvus_demo.db")]
    r1 = subprocess.run([sys.executable, "-m", "astts.cli.search_json"] + args + ["--output_file", str(tmp_path / "one.jsonl")], env=base_env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", str(free_port()), "-m", "astts.cli.search_json"] + args + ["--output_file", str(tmp_path / "dist.jsonl")],
                        env=dict(base_env, ASTTS_FORCE_DIST="1"), capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    one, dist = (tmp_path / "one.jsonl").read_bytes(), (tmp_path / "dist.jsonl").read_bytes()
    assert one == dist and one.count(b"\n") == 37
    recs = [json.loads(l) for l in one.decode().splitlines()]
    meta = json.load(open(os.path.join(gold, "style_bank_meta.json")))
    assert [r["retrieved_file_id"] for r in recs] == [meta["rows"][int(i)]["file_id"] for i in rows]        # 0.3 sigma of noise: top-1 is the source row


def _clean_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "ASTTS_BENCH_STUB", "ASTTS_BENCH_FORCE_DIST"):
        if k not in extra:
            env.pop(k, None)
    return env


def _roofline_ok(r):
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0.0 < r["frac"] < 1.0, r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["avg_us"] > 0


@pytest.mark.gpu
def test_bench_side_workload_as_a_line_of_its_own():
    """`python bench.py --workload config3` (BASELINE configs[2] as a line of its own): one JSON line on stdout with the default line's
    keys, `side_measurement` set, a finite waveform, its own dominant-kernel roofline and the per-kind rooflines."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config3", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in res, key
    assert res["side_measurement"].startswith("--workload config3") and "workload" in res["config"] and res["waveform_finite"] is True
    assert res["value"] > 50.0 and res["n_gpus"] == 1 and res["steps"] == 1
    _roofline_ok(res["roofline"])
    assert set(res["roofline_by_kind"]) == {"gemm_tile", "lm_gemv", "attn_mha_flash", "lm_attn"}
    print("bench --workload config3:", round(res["value"], 1), "x real time,", round(res["ms_per_step"], 1), "ms per step;",
          res["roofline"]["kernel"], round(res["roofline"]["frac"], 3))


@pytest.mark.gpu
def test_default_bench_line_carries_every_baseline_config():
    """The DEFAULT line (`python bench.py --gpus 1`, what the driver times) carries bounded passes of BASELINE configs[2..4] under
    `side_workloads` -- each with value, ms_per_step, its dominant kernel's roofline and per-kind rooflines -- and the 100k-bank
    retrieval stress under `knn_stress` (Q = 8 against the HBM roofline, Q = 256 against the MFMA roofline, ids checked on a sample)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-24khz",
                        "--no-cobatch"], env=_clean_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["ids_match_oracle"] is True and res["waveform_finite_and_clamped"] is True and res["value"] > 50.0
    _roofline_ok(res["roofline"])
    side = res["side_workloads"]
    assert set(side) == {"config3", "config4", "config5"}
    for name, s in side.items():
        assert s["value"] > 50.0 and s["ms_per_step"] > 0 and s["waveform_finite"] is True, (name, s)
        _roofline_ok(s["roofline"])
        assert set(s["roofline_by_kind"]) == {"gemm_tile", "lm_gemv", "attn_mha_flash", "lm_attn"}
    assert side["config4"]["ids_match_oracle_sample"] is True and side["config5"]["ids_match_oracle_sample"] is True
    assert side["config3"]["scaling"] == "weak" and side["config4"]["scaling"] == "strong" and side["config5"]["scaling"] == "strong"
    ks = res["knn_stress"]
    assert set(ks) == {"N100000_D6144_Q8", "N100000_D6144_Q256", "N100000_D768_Q256"}
    for name, k in ks.items():
        assert k["ids_match_oracle_sample"] is True and k["qps"] > 0, (name, k)
        _roofline_ok(k["roofline"])
    assert ks["N100000_D6144_Q8"]["roofline"]["bound"] == "hbm" and ks["N100000_D6144_Q256"]["roofline"]["bound"] == "mfma"
    # round 6: the embedder at its real depth (measured, not extrapolated), the host-I/O-inclusive number with the GPU frontend inside,
    # streaming statistics with the 0.9 bar
    emb = res["embedder"]
    assert emb.get("error") is None and emb["layers"] == 28 and emb["vocab"] == 128256 and emb["texts_per_s"] > 0 and emb["finite"] is True
    assert not any(k.startswith("extrapolated") for k in emb)
    assert res["value_with_host_io"] > 50.0 and res["value_with_host_io"] < res["value"] and res["frontend_ms"] > 0
    assert res["host_io"]["frontend"]["speech_tokenizer"] == "synthetic-weights" and res["host_io"]["frontend"]["speaker_embedder"] == "synthetic-weights"
    st = res["streaming"]
    assert st.get("error") is None and st["tokens_250"]["trials"] >= 5
    assert st["tokens_250"]["live_first_chunk_below_0p9_of_one_pass"] is True and st["tokens_500"]["live_first_chunk_below_0p9_of_one_pass"] is True
    assert res.get("cpu_baseline") is None or "ONE utterance" in res["cpu_baseline"]["unit"]
    print("default line:", round(res["value"], 1), "x;", {n: round(s["value"], 1) for n, s in side.items()},
          {n: (round(k["qps"]), round(k["roofline"]["frac"], 3)) for n, k in ks.items()})
