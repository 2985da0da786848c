#!/bin/bash
# wgrad kernel alone at the configs[1] layer shapes, in-tree library against timing-only ablation builds (scripts/build_variant_one.sh
# wnosplit conv_wgrad_halo.hip -DSGG_WABL_NOSPLIT=1; wnostage ... -DSGG_WABL_NOSTAGE=1): bash scripts/gpu_wgrad_abl.sh <tag> base wnosplit ...
set -e
TAG=$1; shift
mkdir -p gpurun_out/$TAG
SHAPES=("64 224 32 32 3 1" "64 112 32 64 3 1" "64 112 64 64 3 1" "64 112 64 128 3 1" "64 112 128 128 3 1" "64 56 128 256 3 1" "64 56 256 256 3 1" "64 112 128 128 5 2" "64 56 256 512 5 2" "64 28 512 512 5 2")
for v in "$@"; do
  if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
  for sh in "${SHAPES[@]}"; do
    echo -n "[$v] " | tee -a gpurun_out/$TAG/wgrad.log
    timeout -k 10 120 python scripts/prof_conv.py $sh 20 ${MODE:-wgrad} 2>&1 | tail -1 | tee -a gpurun_out/$TAG/wgrad.log
  done
done
