#!/bin/bash
# conv1_1 forward alone (batch 64, 224 x 224) compiled for 4 / 5 / 6 resident workgroups per CU; kernel + step tests on the default
set -e
mkdir -p gpurun_out/c3abl
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -x > gpurun_out/c3abl/pytest.log 2>&1 || { tail -40 gpurun_out/c3abl/pytest.log; exit 1; }
tail -2 gpurun_out/c3abl/pytest.log
for v in base c3wg5 c3wg6; do
  if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
  for rep in 1 2; do
    echo -n "[$v] "; python scripts/prof_conv.py 64 224 3 32 3 1 20 fwd
    echo -n "[$v] "; python scripts/prof_conv.py 64 224 3 32 3 1 20 fwd_stats
  done
done
