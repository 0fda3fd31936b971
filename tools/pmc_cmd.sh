#!/bin/bash
# PMC passes (separate, as the microarchitecture guide prescribes) over any python command:
#   NFDP_PMC_KERNEL_RE='regex' tools/pmc_cmd.sh <tag> <script and args...>   -> gpurun_out/<tag>_pmc.json
set -e -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
  echo "pmc pass $name done"
done
python3 tools/pmc_summary.py "$PWD/gpurun_out/${TAG}_pmc.json" "$OUT"/FETCH_SIZE "$OUT"/WRITE_SIZE "$OUT"/SQ_WAVE_CYCLES "$OUT"/TCC_HIT_sum "$OUT"/GRBM_GUI_ACTIVE > /dev/null
rm -rf "$OUT"
cat "$PWD/gpurun_out/${TAG}_pmc.json"
