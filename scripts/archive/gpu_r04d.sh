#!/bin/bash
# round 4, call d: LDS-DMA filter gradient + full-line pre-split stores: tests, whole suite, A/B, rocprofv3 kernel statistics of the
# serial step with presplit on / off
set -e
mkdir -p gpurun_out/r04d
timeout -k 10 600 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x > gpurun_out/r04d/pytest_presplit.log 2>&1 || { tail -40 gpurun_out/r04d/pytest_presplit.log; exit 1; }
tail -2 gpurun_out/r04d/pytest_presplit.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04d/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04d/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r04d/pytest_gpu.log
bash scripts/gpu_opt_ab.sh r04d_ab "" "presplit=0" "ln_fusion_skip_bwd=4+5+6"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in on off; do
  if [ $v = off ]; then export SGG_OPTIONS="presplit=0"; else unset SGG_OPTIONS; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04d/stats_$v -o k -- python3 $R/bench.py --steps 5 --warmup 2 --single-stream --serial-steps 0 --other-configs 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --no-kernel-timing > $R/gpurun_out/r04d/bench_rocprof_$v.json 2> $R/gpurun_out/r04d/stats_$v.err
  rm -f $R/gpurun_out/r04d/stats_$v/k_kernel_trace.csv
  find $R/gpurun_out/r04d/stats_$v -name "*kernel_stats.csv" | head -2
done
