#!/bin/bash
# round 4, call j: 256-column (eight-wave) workgroups of the band-resident 5x5 stride-2 kernel: tests, same-box A/B against -DS2_WIDE=0,
# per-call rates of the serial steps, LN6 prologue off in passes with a backward
set -e
mkdir -p gpurun_out/r04j
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_presplit_gpu.py tests/test_step_gpu.py tests/test_configs34_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x > gpurun_out/r04j/pytest.log 2>&1 || { tail -40 gpurun_out/r04j/pytest.log; exit 1; }
tail -2 gpurun_out/r04j/pytest.log
bash scripts/gpu_ab.sh r04j_ab base nowide
for v in base nowide; do
  if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 --per-shape > gpurun_out/r04j/serial_$v.json 2> gpurun_out/r04j/serial_$v.err
done
unset SGG_HIP_LIB
bash scripts/gpu_opt_ab.sh r04j_opt "" "ln_fusion_skip_bwd=6"
