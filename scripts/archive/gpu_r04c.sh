#!/bin/bash
# round 4, call c: pre-split activations - kernel tests, the whole GPU suite, same-box A/B of the full step against presplit=0
set -e
mkdir -p gpurun_out/r04c
timeout -k 10 600 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x > gpurun_out/r04c/pytest_presplit.log 2>&1 || { tail -40 gpurun_out/r04c/pytest_presplit.log; exit 1; }
tail -2 gpurun_out/r04c/pytest_presplit.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04c/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r04c/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r04c/pytest_gpu.log
bash scripts/gpu_opt_ab.sh r04c_ab "" "presplit=0"
echo "== GPU-bound per-shape head GEMM timings ==" 
timeout -k 10 200 python scripts/prof_gemm.py 2>&1 | tee gpurun_out/r04c/prof_gemm_new.log | tail -22
SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_nowk.so timeout -k 10 200 python scripts/prof_gemm.py 2>&1 | tee gpurun_out/r04c/prof_gemm_old.log | tail -22
