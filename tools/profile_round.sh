#!/bin/bash
# Round-end profiling on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command (eager launches so every kernel is a dispatch)
#   2. separate --pmc passes over the fused flow-step kernels (tools/time_fused_level.py: one persistent launch per level,
#      16 distinct flows each)
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries into profiles/.
set -e -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-graph > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
echo "stats pass done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  NFDP_LEVEL_ONLY=1 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 tools/time_fused_level.py > "$OUT/pmc_$name.log" 2>&1
  echo "pmc pass $name done"
done
python3 tools/pmc_summary.py "$OUT/pmc_summary.json" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_SQ_WAVE_CYCLES "$OUT"/pmc_TCC_HIT_sum > /dev/null
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
ls "$OUT"
# 3. the other configs: kernel stats of the CelebA-shape forward / inverse (tools/bench_celeba.py) and of the training step
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/celeba" -o celeba -- python3 tools/bench_celeba.py > "$OUT/celeba_under_rocprof.txt" 2>&1 || true
find "$OUT/celeba" -name "*kernel_stats.csv" -exec cp {} "$OUT/celeba_kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train" -o train -- python3 bench.py --train --steps 5 --warmup 2 --no-graph > "$OUT/train_under_rocprof.json" 2> "$OUT/train_under_rocprof.err" || true
find "$OUT/train" -name "*kernel_stats.csv" -exec cp {} "$OUT/train_kernel_stats.csv" \;
ls "$OUT"
# 4. the fp32 mode with split products (32x32 L3 K16 B = 512 forward, eager launches)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f32x3" -o f32x3 -- python3 tools/trace_glow_fwd.py 32 3 512 f32 bf16x3 > "$OUT/f32x3_under_rocprof.txt" 2>&1 || true
find "$OUT/f32x3" -name "*kernel_stats.csv" -exec cp {} "$OUT/f32x3_kernel_stats.csv" \;
ls "$OUT"
