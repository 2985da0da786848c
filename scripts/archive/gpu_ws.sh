# wide-store epilogue + head side stream: the kernel / full-size / step / concurrency tests FIRST (a wrong kernel is never timed), then
# per-shape timings against the previous library (scripts/build_old_lib.sh HEAD) and the same-box A/B in the full step.
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-ws}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_conv_gpu.py tests/test_step_gpu.py tests/test_configs34_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
  for v in old base; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    for spec in "64 112 128 128 3 1 fwd_ws" "64 56 256 256 3 1 fwd_ws" "64 224 32 32 3 1 fwd_ws" "64 112 64 64 3 1 fwd_ws" "64 112 128 128 3 1 dgrad_ws" "64 112 128 128 5 2 fwd_ws" "64 56 256 512 5 2 fwd_ws" "64 56 256 512 5 2 dgrad_ws"; do
      set -- $spec
      echo -n "$v rep $rep: " | tee -a $O/times.log
      timeout -k 10 120 python scripts/prof_conv.py $1 $2 $3 $4 $5 $6 20 $7 2>&1 | tail -1 | tee -a $O/times.log
    done
  done
done
unset SGG_HIP_LIB
bash scripts/gpu_ab.sh ${1:-ws}_ab old base
