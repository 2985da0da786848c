set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_step_gpu.py tests/test_golden.py tests/test_api_gpu.py -m gpu -q 2>&1 | tail -2
bash scripts/gpu_bench_short.sh
SGG_CONV_PRECISION=3 bash scripts/gpu_bench_short.sh
