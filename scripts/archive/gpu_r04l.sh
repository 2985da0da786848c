#!/bin/bash
# round 4, call l: head parameter-gradient work deferred to a side stream behind one fork per pass: concurrency + step tests, A/B
set -e
mkdir -p gpurun_out/r04l
timeout -k 10 900 python -m pytest tests/test_concurrency_gpu.py tests/test_step_gpu.py tests/test_presplit_gpu.py tests/test_api_gpu.py -m gpu -q -x > gpurun_out/r04l/pytest.log 2>&1 || { tail -40 gpurun_out/r04l/pytest.log; exit 1; }
tail -2 gpurun_out/r04l/pytest.log
bash scripts/gpu_flag_ab.sh r04l_ab "--head-side-stream 0" "--head-side-stream 1"
