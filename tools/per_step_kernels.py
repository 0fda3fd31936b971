#!/usr/bin/env python3
"""Kernels (and runtime copy / fill kernels) PER STEP of a bench.py mode, with the process's one-off work subtracted: the same command
is run twice under `rocprofv3 --kernel-trace --stats` with two step counts and the per-kernel call counts / durations are differenced.

    python3 tools/per_step_kernels.py OUTDIR N1 N2 -- --train --no-graph          # bench.py arguments after --

(rocprofv3's kernel_stats of ONE run folds model construction - ~600 parameter uploads - and warm-up into the totals: dividing those by
the step count is how round 4 arrived at "362 copyBuffer launches per training step".)"""
import csv, glob, os, subprocess, sys

out, n1, n2 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
extra = sys.argv[sys.argv.index("--") + 1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, TMPDIR="/tmp", NFDP_BENCH_FIXED_WARMUP="0")       # (bench.py: no time-based clock warm-up - fixed launch counts)


def run(n):
    d = os.path.join(out, f"n{n}")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "t", "--",
           sys.executable, os.path.join(root, "bench.py"), "--steps", str(n), "--warmup", "2"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(r.returncode)
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = {r_["Name"]: (int(r_["Calls"]), float(r_["TotalDurationNs"])) for r_ in csv.DictReader(open(f))}
    for big in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(big)                                             # (tens of MB; the stats file is what is kept)
    return rows


a, b = run(n1), run(n2)
rows = []
for k in set(a) | set(b):
    c1, t1 = a.get(k, (0, 0.0)); c2, t2 = b.get(k, (0, 0.0))
    rows.append(((t2 - t1) / (n2 - n1), (c2 - c1) / (n2 - n1), c1 - (c2 - c1) / (n2 - n1) * (n1 + 2), k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"per step (difference of {n2}- and {n1}-step runs): {tot / 1e3:.1f} us of kernel time, {sum(r[1] for r in rows):.1f} launches")
print(f"{'us/step':>10} {'calls/step':>10} {'one-off calls':>13}  kernel")
for t, c, once, k in rows:
    if abs(c) > 1e-9 or abs(once) > 0.5:
        print(f"{t / 1e3:10.1f} {c:10.2f} {once:13.0f}  {k[:150]}")
