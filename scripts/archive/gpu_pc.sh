# producer / consumer halo kernel: correctness first (short timeouts: a barrier mismatch would hang), then timings against the
# same library without it (-DSGG_HALO_PC=0) on the same box.
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-pc}
mkdir -p $O
timeout -k 5 150 python -m pytest tests/test_kernels_gpu.py -k "test_conv_halo_fwd_dgrad" -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
timeout -k 5 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $O/t2.log 2>&1 || { tail -40 $O/t2.log; exit 1; }
tail -2 $O/t2.log
timeout -k 10 600 python -m pytest tests/test_fullsize_conv_gpu.py tests/test_step_gpu.py tests/test_configs34_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x > $O/t3.log 2>&1 || { tail -40 $O/t3.log; exit 1; }
tail -2 $O/t3.log
for rep in 1 2; do
  for v in nopc base; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    for spec in "64 112 128 128 3 1 fwd_ws" "64 56 256 256 3 1 fwd_ws" "64 112 128 128 3 1 dgrad_ws" "64 112 64 128 3 1 fwd_ws" "64 56 128 256 3 1 fwd_ws" "64 112 128 128 3 1 fwd_ws_ln"; do
      set -- $spec
      echo -n "$v rep $rep: " | tee -a $O/times.log
      timeout -k 10 120 python scripts/prof_conv.py $1 $2 $3 $4 $5 $6 20 $7 2>&1 | tail -1 | tee -a $O/times.log
    done
  done
done
unset SGG_HIP_LIB
bash scripts/gpu_ab.sh ${1:-pc}_ab nopc base
