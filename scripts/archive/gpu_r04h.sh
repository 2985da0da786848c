#!/bin/bash
# round 4, call h: band-resident 5x5 stride-2 kernel with the patch staged by LDS-DMA: tests, same-box A/B against -DS2_DMA=0
set -e
mkdir -p gpurun_out/r04h
true
true
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_configs34_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x > gpurun_out/r04h/pytest_step.log 2>&1 || { tail -40 gpurun_out/r04h/pytest_step.log; exit 1; }
tail -2 gpurun_out/r04h/pytest_step.log
bash scripts/gpu_ab.sh r04h_ab base nos2dma
