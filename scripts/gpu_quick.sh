set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q 2>&1 | tail -2
for args in "64 112 128 128 3 1 5 wgrad" "64 56 256 512 5 2 5 wgrad" "64 28 512 512 5 2 5 wgrad" "64 224 32 32 3 1 5 wgrad" "64 224 32 32 5 2 5 wgrad"; do
  timeout -k 10 120 python scripts/prof_conv.py $args 2>/dev/null
done
bash scripts/gpu_bench_short.sh
