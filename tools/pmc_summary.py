#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into per-kernel averages per launch.

usage: python tools/pmc_summary.py OUT.json DIR [DIR...]      (NFDP_PMC_KERNEL_RE: regex of the kernels to keep, default the fused
flow-step kernels)
Each DIR is the -d directory of one `rocprofv3 --pmc ... -- python3 tools/time_fused_level.py` pass (separate passes:
FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  HBM bytes per launch =
2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B, same guide, 'HBM') + WRITE_SIZE; both counters are in KB.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))          # kernel -> counter -> [sum, dispatches]
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])     # a counter is reported per XCD/instance: sum them
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, ctr), v in per_dispatch.items():
            m = re.search(os.environ.get("NFDP_PMC_KERNEL_RE", r"coupling_fused\w*kernel<[^>]*>"), names[disp])
            if not m:
                continue
            a = acc[m.group(0)][ctr]
            a[0] += v
            a[1] += 1
res = {}
for k, ctrs in sorted(acc.items()):
    e = {c: round(v[0] / v[1], 2) for c, v in sorted(ctrs.items())}
    e["launches_averaged"] = max(v[1] for v in ctrs.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        for c, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"),
                        ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac")):
            if c in e:
                e[name] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            # the MFMA-busy counter advances 4 per busy cycle of a SIMD (it reads 4x the wave-cycles of a kernel whose one
            # wave per SIMD issues MFMAs back to back), hence the /4: fraction of a wave's lifetime with its MFMA pipe busy
            e["mfma_busy_over_wave_cycles"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / e["SQ_WAVE_CYCLES"], 4)
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
        e["TCC_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
    res[k] = e
# provenance: the kernel sources and the commit the counters were collected on (bench.py reports whether they are still current)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import kernel_source_sha16
    head = None
    try:
        import subprocess
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        pass
    res["_meta"] = {"kernel_source_sha16": kernel_source_sha16(), "git_head": head or os.environ.get("NFDP_GIT_HEAD")}
except Exception as e:      # noqa: BLE001
    res["_meta"] = {"error": str(e)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
