set -e
cd $GRAFT_REPO_ROOT
SGG_GATHER_WIDE=1 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv and f16x3" 2>&1 | tail -2
for v in 0 1; do
echo "wide=$v"
for args in "64 56 256 256 3 1 5 fwd" "64 56 256 512 5 2 5 fwd" "64 28 512 512 5 2 5 fwd" "64 56 256 256 3 1 5 dgrad"; do
  SGG_GATHER_WIDE=$v timeout -k 10 120 python scripts/prof_conv.py $args 2>/dev/null
done
done
