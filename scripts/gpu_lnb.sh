# one-pass LayerNorm backward (build it with scripts/build_variant_one.sh lnf layernorm.hip -DSGG_LN_BWD_FUSED=1): growing sizes (each its own process and time limit), tests, the streaming kernels
# alone, then the whole step against the two-pass build
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-lnb}
mkdir -p $O
for spec in "2 40 40 64" "2 112 112 128" "8 112 112 128" "64 112 112 128" "64 224 224 32" "64 56 56 256" "64 14 14 512"; do
  timeout -k 5 90 python scripts/check_ln_bwd_scale.py $spec >> $O/scale.log 2>&1 || { tail -5 $O/scale.log; exit 1; }
  tail -1 $O/scale.log
done
timeout -k 10 240 python -m pytest tests/test_kernels_gpu.py -k "layernorm or ln_" -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
for spec in "64 112 112 128" "64 224 224 32" "64 56 56 256"; do
  SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_lnf.so timeout -k 5 90 python scripts/check_ln_bwd_scale.py $spec >> $O/scale.log 2>&1 || { tail -5 $O/scale.log; exit 1; }
  tail -1 $O/scale.log
done
timeout -k 10 500 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -m gpu -q -x > $O/t3.log 2>&1 || { tail -40 $O/t3.log; exit 1; }
tail -2 $O/t3.log
for rep in 1 2; do
  for v in lnf base; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 0 > $O/$v.$rep.json 2> $O/$v.$rep.err
    python - <<PY | tee -a $O/ab.log
import json
d=json.loads(open('$O/$v.$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: %.2f ms/step  %.1f triples/s  parity %s' % (d['ms_per_step'], d['value'], d.get('parity', {}).get('ok')))
PY
  done
done
