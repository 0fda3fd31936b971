#!/bin/bash
# A/B of the level-0 launch: 4 waves x 64 px against 8 waves x 32 px, over the ablation builds in nfdp_hip/libnfdp_hip_<tag>.so
P=$PWD/normalizing-flow-with-diffusion-prior-model_amd/nfdp_hip
for tag in "" $@; do
  for nw in 4 8; do
    lib=$P/libnfdp_hip${tag:+_$tag}.so
    r=$(NFDP_LIB_PATH=$lib NFDP_L0_WAVES=$nw NFDP_LEVEL_ONLY=1 timeout -k 10 120 python tools/time_fused_level.py 12 16 2>&1 | grep "level launch")
    echo "build=${tag:-prod} waves=$nw same_w=${NFDP_SAME_W:-0} $r"
  done
done
