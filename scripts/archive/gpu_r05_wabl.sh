# round 5: where conv1_2's / conv1_3's filter gradients (the last matrix kernels of every backward) spend their time: timing-only ablations
set -e
mkdir -p gpurun_out/r05_wabl
for v in base wnosplit wnostage; do
  if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
  for m in wgrad wgrad_ln; do
    for sh in "64 224 32 32 3 1" "64 224 32 32 5 2" "64 112 32 64 3 1"; do
      echo -n "[$v] " | tee -a gpurun_out/r05_wabl/wgrad.log
      timeout -k 10 120 python scripts/prof_conv.py $sh 20 $m 2>&1 | tail -1 | tee -a gpurun_out/r05_wabl/wgrad.log
    done
  done
done
