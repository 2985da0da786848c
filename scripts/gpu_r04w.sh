#!/bin/bash
# round 4, call w: option d_early (D's encoder forward of the next critic update beside the generator update's heads and backward)
set -e
mkdir -p gpurun_out/r04w
SGG_OPTIONS="d_early=1" timeout -k 10 600 python -m pytest tests/test_concurrency_gpu.py -m gpu -q -x > gpurun_out/r04w/pytest.log 2>&1 || { tail -40 gpurun_out/r04w/pytest.log; exit 1; }
tail -2 gpurun_out/r04w/pytest.log
bash scripts/gpu_opt_ab.sh r04w_opt "" "d_early=1"
