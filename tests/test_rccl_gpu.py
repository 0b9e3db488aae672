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
                        "--no-cpu-baseline", "--no-24khz"], env=env, capture_output=True, text=True, timeout=900)
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
                        "--no-cpu-baseline", "--no-24khz", "--no-cobatch"], env=env, capture_output=True, text=True, timeout=900)
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


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["config3", "config5"])
def test_bench_side_workloads_print_the_same_schema(workload):
    """`python bench.py --workload config3|config5` (BASELINE configs[2] / configs[4] as side lines): one JSON line on stdout with the
    default line's keys, `side_measurement` set, a finite waveform and -- config 5 -- retrieved ids equal to the oracle's on a sample."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "ASTTS_BENCH_STUB", "ASTTS_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in res, key
    assert res["side_measurement"].startswith(f"--workload {workload}") and "workload" in res["config"] and res["waveform_finite"] is True
    assert res["value"] > 50.0 and res["n_gpus"] == 1 and res["steps"] == 1
    if workload == "config5":
        assert res["ids_match_oracle_sample"] is True and res["scaling"] == "strong"
    print(f"bench --workload {workload}:", round(res["value"], 1), "x real time,", round(res["ms_per_step"], 1), "ms per step")
