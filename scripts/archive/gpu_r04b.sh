#!/bin/bash
# round 4, call b: the workgroup-split-K head GEMM (tests, per-shape A/B against the slab kernels, full step A/B), wgrad ablations
set -e
mkdir -p gpurun_out/r04b
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or lnlstm or attention" > gpurun_out/r04b/pytest_gemm.log 2>&1 || { tail -30 gpurun_out/r04b/pytest_gemm.log; exit 1; }
tail -2 gpurun_out/r04b/pytest_gemm.log
echo "== new (in-tree) ==" | tee gpurun_out/r04b/prof_gemm.log
timeout -k 10 200 python scripts/prof_gemm.py 2>&1 | tee -a gpurun_out/r04b/prof_gemm.log
echo "== old (SGG_GEMM_WK=0) ==" | tee -a gpurun_out/r04b/prof_gemm.log
SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_nowk.so timeout -k 10 200 python scripts/prof_gemm.py 2>&1 | tee -a gpurun_out/r04b/prof_gemm.log
timeout -k 10 600 python -m pytest tests/test_step_gpu.py tests/test_configs34_gpu.py -m gpu -q -x > gpurun_out/r04b/pytest_step.log 2>&1 || { tail -30 gpurun_out/r04b/pytest_step.log; exit 1; }
tail -2 gpurun_out/r04b/pytest_step.log
bash scripts/gpu_ab.sh r04b_ab base nowk
bash scripts/gpu_wgrad_abl.sh r04b base wnosplit wnostage
