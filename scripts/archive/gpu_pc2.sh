# producer / consumer halo kernel, iteration: correctness (halo + LN-prologue + step tests), timings of variants on the same box
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-pc2}
mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_kernels_gpu.py -k "halo or prologue or epilogue or prepare_weights" -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
timeout -k 10 400 python -m pytest tests/test_fullsize_conv_gpu.py tests/test_step_gpu.py -m gpu -q -x > $O/t3.log 2>&1 || { tail -40 $O/t3.log; exit 1; }
tail -2 $O/t3.log
for rep in 1 2; do
  for v in $VARIANTS; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    for spec in "64 112 128 128 3 1 fwd_ws" "64 56 256 256 3 1 fwd_ws" "64 112 128 128 3 1 dgrad_ws" "64 112 128 128 3 1 fwd_ws_ln"; do
      set -- $spec
      echo -n "$v rep $rep: " | tee -a $O/times.log
      timeout -k 10 120 python scripts/prof_conv.py $1 $2 $3 $4 $5 $6 20 $7 2>&1 | tail -1 | tee -a $O/times.log
    done
  done
done
unset SGG_HIP_LIB
