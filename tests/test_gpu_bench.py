"""The driver's contract for bench.py (one JSON line on stdout) and __graft_entry__.smoke(), run as the driver runs them: child
processes of their own."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    env.pop("NFDP_LIB_PATH", None)
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = _run(["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-legs"])
    lines = [ln for ln in out.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3            # images/s of the whole job
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e7
    assert 6.0 < d["bits_per_dim"] < 8.0


def test_graft_entry_smoke_runs_against_the_oracle():
    out = _run(["-c", "import __graft_entry__ as g; g.smoke()"])
    assert "smoke:" in out and "max rel err" in out
