"""`python bench.py --gpus N` must work without torch.distributed.run around it (the driver's command form): the parent
spawns one rank per GPU before any HIP call.  Checked here on CPU with the stub step over gloo (ASTTS_BENCH_STUB=1:
same rendezvous / barrier / max-over-ranks / one-JSON-line protocol, a sleep as the step)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env, *argv, timeout=180):
    env = dict(os.environ, ASTTS_BENCH_STUB="1", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks_prints_one_json_line():
    r = _run({}, "--gpus", "2", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_world_size"] == 2 and res["steps"] == 4 and res["warmup"] == 1
    assert res["ranks_seen"] == [0, 1]
    # max over ranks: rank 1 sleeps 4 ms per step, rank 0 sleeps 2 ms
    assert res["ms_per_step"] >= 3.9


def test_self_launch_propagates_a_failing_rank():
    r = _run({"ASTTS_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "0")
    assert r.returncode != 0


def test_single_rank_needs_no_launcher():
    r = _run({}, "--gpus", "1", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_external_launcher_env_is_respected():
    """Under torch.distributed.run the ranks already exist (RANK set): bench.py must not spawn again."""
    env = dict(os.environ, ASTTS_BENCH_STUB="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_self_launch_eight_ranks_the_node_the_north_star_names():
    """`python bench.py --gpus 8` (the driver's SCALE run): eight ranks rendezvous, barrier, reduce the MAX time and print ONE line."""
    r = _run({}, "--gpus", "8", "--steps", "3", "--warmup", "1", timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["rccl_world_size"] == 8 and res["ranks_seen"] == list(range(8))
    assert res["ms_per_step"] >= 15.9          # the slowest rank (rank 7 sleeps 16 ms per step) sets the time
