#!/bin/bash
# round 4, call n: encoder forward of the untouched network beside the previous update's encoder backward (option cross_step): GPU
# suite, same-box A/B; wgrad_late is the default now
set -e
mkdir -p gpurun_out/r04n
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04n/pytest.log 2>&1 || { tail -40 gpurun_out/r04n/pytest.log; exit 1; }
tail -2 gpurun_out/r04n/pytest.log
bash scripts/gpu_opt_ab.sh r04n_opt "cross_step=0" "" "cross_step=0,wgrad_late=0"
