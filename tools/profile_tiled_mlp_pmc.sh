#!/bin/bash
# PMC passes over the tiled coupling-network kernels (tools/time_mlp_tiled.py); summary -> gpurun_out/pmc_mlp/summary.json (profiles/r04_z_tiled_mlp_counters.json)
set -e
OUT=$PWD/gpurun_out/pmc_mlp
mkdir -p $OUT
export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 tools/time_mlp_tiled.py > "$OUT/pmc_$name.log" 2>&1 || echo "pass $name failed"
  echo "pmc pass $name done"
done
NFDP_PMC_KERNEL_RE='coupling_mlp\w*kernel<[^>]*>' python3 tools/pmc_summary.py "$OUT/summary.json" "$OUT"/pmc_SQ_WAVE_CYCLES "$OUT"/pmc_SQ_WAIT_INST_LDS "$OUT"/pmc_SQ_ACTIVE_INST_LDS > /dev/null
