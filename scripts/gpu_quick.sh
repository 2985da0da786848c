set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" 2>&1 | tail -3
for prec in 0 6 3; do
echo "precision $prec"
for args in "64 112 128 128 3 1 5 wgrad" "64 56 256 512 5 2 5 wgrad" "64 28 512 512 5 2 5 wgrad" "64 224 32 32 3 1 5 wgrad" "64 112 64 64 3 1 5 wgrad"; do
  SGG_CONV_PRECISION=$prec timeout -k 10 120 python scripts/prof_conv.py $args 2>/dev/null
done
done
for prec in 6 3; do SGG_CONV_PRECISION=$prec bash scripts/gpu_bench_short.sh; done
