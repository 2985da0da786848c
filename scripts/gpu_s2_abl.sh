#!/bin/bash
# band-resident 5x5 stride-2 kernel alone at the configs[1] layer shapes (forward with pre-split weights, dgrad), in-tree library
# against timing-only ablation builds: bash scripts/gpu_s2_abl.sh <tag> base s2nostage s2nob s2noepi s2none
set -e
TAG=$1; shift
mkdir -p gpurun_out/$TAG
SHAPES=("64 112 128 128 5 2" "64 56 256 512 5 2" "64 28 512 512 5 2")
for v in "$@"; do
  if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
  for sh in "${SHAPES[@]}"; do
    for m in fwd_ws dgrad_ws; do
      echo -n "[$v] " | tee -a gpurun_out/$TAG/s2.log
      timeout -k 10 120 python scripts/prof_conv.py $sh 20 $m 2>&1 | tail -1 | tee -a gpurun_out/$TAG/s2.log
    done
  done
done
