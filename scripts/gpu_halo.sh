set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "halo" > gpurun_out/halo_tests.log 2>&1 || { tail -30 gpurun_out/halo_tests.log; exit 1; }
tail -2 gpurun_out/halo_tests.log
timeout -k 10 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
bash scripts/gpu_bench_short.sh
