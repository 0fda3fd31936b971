#!/bin/bash
# Kernel durations of one command under rocprofv3 (run through gpurun from the repo root):
#   tools/kprof.sh <tag> <name filter (substring, '' = all)> <python script and args...>
# -> gpurun_out/<tag>_kstats.csv (the whole table) and, on stdout, calls / average / min ns of the kernels whose name contains the filter.
set -e -o pipefail
TAG=$1; FILT=$2; shift 2
ROOT=$PWD
OUT=$ROOT/gpurun_out/kprof_$TAG
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$@" > "$OUT.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/${TAG}_kstats.csv" \;
python3 - "$ROOT/gpurun_out/${TAG}_kstats.csv" "$FILT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f'{int(r["Calls"]):6d} calls  avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f} us  {r["Name"][:110]}')
PY
rm -rf "$OUT"
