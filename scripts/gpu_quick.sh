set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" 2>&1 | tail -2
for prec in 6 3; do
echo "=== precision $prec"
SGG_CONV_PRECISION=$prec timeout -k 10 300 python -m pytest tests/test_step_gpu.py tests/test_golden.py -m gpu -q 2>&1 | tail -4
SGG_CONV_PRECISION=$prec bash scripts/gpu_bench_short.sh
done
