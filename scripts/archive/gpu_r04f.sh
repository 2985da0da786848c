#!/bin/bash
# round 4, call f: row-band LDS-DMA filter gradient, conv1_1 kernels with register prefetch, LN-prologue plan for pre-split consumers
set -e
mkdir -p gpurun_out/r04f
timeout -k 10 600 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x > gpurun_out/r04f/pytest_presplit.log 2>&1 || { tail -40 gpurun_out/r04f/pytest_presplit.log; exit 1; }
tail -2 gpurun_out/r04f/pytest_presplit.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04f/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04f/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r04f/pytest_gpu.log
bash scripts/gpu_opt_ab.sh r04f_ab "" "presplit=0"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04f/stats -o k -- python3 $R/bench.py --steps 5 --warmup 2 --single-stream --serial-steps 0 --other-configs 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --no-kernel-timing > $R/gpurun_out/r04f/bench_rocprof.json 2> $R/gpurun_out/r04f/stats.err
rm -f $R/gpurun_out/r04f/stats/k_kernel_trace.csv
