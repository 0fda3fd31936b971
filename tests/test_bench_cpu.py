"""bench.py's multi-rank plumbing on a CPU-only host: `python bench.py --gpus N` with no launcher around it starts its own N ranks
as child processes (before anything touches the GPU), relays rank 0's JSON line and reports a failing rank through its exit status.
`--launch-check` stops after the rendezvous + one all-reduce, so the path runs here over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NFDP_DIST_BACKEND="gloo", **kw)
    return env


def test_self_launch_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert 1024 < bench.free_port() < 65536


def test_bench_gpus_2_launches_its_own_ranks_and_relays_rank0():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line: rank 0's
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "world": 2, "all_reduce_sum": 2, "backend": "gloo"}
    assert "starting 2 ranks" in r.stderr


def test_bench_self_launch_reports_a_failing_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(NFDP_LAUNCH_CHECK_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0


def test_bench_under_a_launcher_does_not_launch_again():
    """Under torch.distributed.run (WORLD_SIZE set) the script is a rank: a mismatch between --gpus and the world is an error."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert "starting 2 ranks" not in r.stderr
