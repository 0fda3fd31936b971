#!/bin/bash
# PMC passes (separate, as the microarchitecture guide prescribes) over one weight-gradient GEMM: tools/profile_wgrad_pmc.sh [dw2 16 12]
set -e -o pipefail
OUT=$PWD/gpurun_out/prof_wgrad_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/$name" -- python3 tools/bench_wgrad.py "$@" > "$OUT/$name.log" 2>&1
  echo "pmc pass $name done"
done
export NFDP_PMC_KERNEL_RE='wgrad_bf16_wide_kernel<[^>]*>|wgrad_bf16_dma_kernel|wgrad_reduce_rows\w*|wgrad_reduce_vec_kernel'
python3 tools/pmc_summary.py "$OUT/wgrad_pmc_summary.json" "$OUT"/FETCH_SIZE "$OUT"/WRITE_SIZE "$OUT"/SQ_WAVE_CYCLES "$OUT"/TCC_HIT_sum "$OUT"/GRBM_GUI_ACTIVE > /dev/null
cat "$OUT/wgrad_pmc_summary.json"
