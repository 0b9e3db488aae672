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
