set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "halo" 2>&1 | tail -1
timeout -k 10 120 python scripts/prof_halo.py 64 112 128 128 10
timeout -k 10 120 python scripts/prof_halo.py 64 224 32 32 10
for shape in "64 112 128 128" "64 56 256 256" "64 112 64 64" "64 224 32 32"; do
  timeout -k 10 120 python scripts/prof_conv.py $shape 3 1 10 fwd_ws
done
