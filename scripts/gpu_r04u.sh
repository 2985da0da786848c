#!/bin/bash
# round 4, call u: option g_early (the generator update's encoder forward beside the preceding critic update's heads and backward)
set -e
mkdir -p gpurun_out/r04u
SGG_OPTIONS="g_early=1" timeout -k 10 600 python -m pytest tests/test_concurrency_gpu.py -m gpu -q -x > gpurun_out/r04u/pytest.log 2>&1 || { tail -40 gpurun_out/r04u/pytest.log; exit 1; }
tail -2 gpurun_out/r04u/pytest.log
bash scripts/gpu_opt_ab.sh r04u_opt "" "g_early=1"
