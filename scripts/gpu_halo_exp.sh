set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "halo" > gpurun_out/halo_tests.log 2>&1 || { tail -40 gpurun_out/halo_tests.log; exit 1; }
tail -2 gpurun_out/halo_tests.log
for shape in "64 112 128 128" "64 56 256 256" "64 56 128 256" "64 112 64 128" "64 112 64 64" "64 112 32 64" "64 224 32 32"; do
  timeout -k 10 120 python scripts/prof_conv.py $shape 3 1 10 fwd_ws
done
