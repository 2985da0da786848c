# round 3, experiment 2: (1) the WB2 tiling (four waves x two blocks x 32 columns) through the kernel / full-size / step tests, then
# (2) per-shape timings of the variants on one box (interleaved repetitions), (3) head side stream on / off in the full step.
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-exp2}
mkdir -p $O
SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_wb2.so timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_conv_gpu.py tests/test_step_gpu.py tests/test_configs34_gpu.py -m gpu -q -x > $O/tests_wb2.log 2>&1 || { tail -40 $O/tests_wb2.log; exit 1; }
tail -2 $O/tests_wb2.log
for rep in 1 2; do
  for v in base wb2 ntst0 nostore nostats wb2nob wb2noepi; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    for spec in "64 112 128 128 3 1 fwd_ws" "64 56 256 256 3 1 fwd_ws" "64 112 128 128 3 1 dgrad_ws"; do
      set -- $spec
      echo -n "$v rep $rep: " | tee -a $O/times.log
      timeout -k 10 120 python scripts/prof_conv.py $1 $2 $3 $4 $5 $6 20 $7 2>&1 | tail -1 | tee -a $O/times.log
    done
  done
done
unset SGG_HIP_LIB
for rep in 1 2; do
  for v in head_single head_side; do
    if [ "$v" = head_single ]; then F="--head-single-stream"; else F=""; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 0 --no-kernel-timing $F > $O/$v.$rep.json 2> $O/$v.$rep.err
    python -c "import json; d=json.loads(open('$O/$v.$rep.json').read().strip().splitlines()[-1]); print('$v rep $rep: %.2f ms/step  %.1f triples/s' % (d['ms_per_step'], d['value']))" | tee -a $O/head.log
  done
done
