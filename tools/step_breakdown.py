#!/usr/bin/env python3
"""Per-kernel time of the LAST traced step in a rocprofv3 --kernel-trace csv (steps are delimited by adam_step_kernel).
usage: python tools/step_breakdown.py <kernel_trace.csv> [top]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_step' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1] + 1]
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:90]
agg = defaultdict(lambda: [0, 0.0])
for r in seg:
    a = agg[short(r['Kernel_Name'])]
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(a[1] for a in agg.values())
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
print(f"last step: {len(seg)} launches, kernel time {tot:.3f} ms, span {span:.3f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{a[1]:7.3f} ms x{a[0]:4d}  {k}")
