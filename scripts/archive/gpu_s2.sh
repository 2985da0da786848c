# s2 kernel: correctness, then per-layer timing against the gather kernel (same box)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "s2" > gpurun_out/s2/test.log 2>&1 || { tail -30 gpurun_out/s2/test.log; exit 1; }
tail -3 gpurun_out/s2/test.log
for shape in "64 112 128 128" "64 56 256 512" "64 28 512 512"; do
  for mode in fwd_ws fwd_gather_ws dgrad_ws dgrad_gather_ws; do
    timeout -k 10 120 python scripts/prof_conv.py $shape 5 2 10 $mode
  done
done 2>&1 | tee gpurun_out/s2/perf.log
