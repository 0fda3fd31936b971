#!/bin/bash
# PMC passes (separate, as the microarchitecture guide prescribes) over one bf16 training step and one CelebA-shape forward:
# MFMA-busy, HBM-side traffic and L2 hit rate of the GEMM / streaming kernels of the training path and of coupling_mlp_kernel.
set -e -o pipefail
OUT=$PWD/gpurun_out/prof_train_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/train_$name" -- python3 tools/bench_train.py 512 1 16 bf16 > "$OUT/train_$name.log" 2>&1
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/celeba_$name" -- python3 tools/trace_celeba_fwd.py > "$OUT/celeba_$name.log" 2>&1
  echo "pmc pass $name done"
done
export NFDP_PMC_KERNEL_RE='wgrad_bf16_multi_kernel|wgrad_reduce_multi_kernel|coupling_bwd_fused_kernel<[^>]*>|coupling_bwd_rs_kernel<[^>]*>|coupling_fused_kernel<[^>]*>|coupling_fused_rs_kernel<[^>]*>|flow_link_bwd_kernel<[^>]*>|coupling_mlp_kernel<[^>]*>|pw_gemm_bf16_kernel|tap_shift_sum_kernel'
python3 tools/pmc_summary.py "$OUT/train_pmc_summary.json" "$OUT"/train_FETCH_SIZE "$OUT"/train_WRITE_SIZE "$OUT"/train_SQ_WAVE_CYCLES "$OUT"/train_TCC_HIT_sum > /dev/null
python3 tools/pmc_summary.py "$OUT/celeba_pmc_summary.json" "$OUT"/celeba_FETCH_SIZE "$OUT"/celeba_WRITE_SIZE "$OUT"/celeba_SQ_WAVE_CYCLES "$OUT"/celeba_TCC_HIT_sum > /dev/null
ls "$OUT"/*.json
