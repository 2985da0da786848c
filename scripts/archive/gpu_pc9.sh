# forward of LN-prologue layers on the four-wave kernel (default) against the producer / consumer kernel for every launch
# (SGG_HALO_PC_LNP=1): step / concurrency tests, then the whole step twice each with the per-kernel rates of the extra steps
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-pc9}
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_fullsize_conv_gpu.py -m gpu -q -x > $O/t3.log 2>&1 || { tail -40 $O/t3.log; exit 1; }
tail -2 $O/t3.log
for rep in 1 2; do
  for v in pcall base; do
    if [ "$v" = base ]; then unset SGG_HALO_PC_LNP; else export SGG_HALO_PC_LNP=1; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 0 > $O/$v.$rep.json 2> $O/$v.$rep.err
    python - <<PY | tee -a $O/ab.log
import json
d=json.loads(open('$O/$v.$rep.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$v rep $rep: %.2f ms/step  %.1f triples/s  dominant %s %.1f TF frac %.3f share %.3f' % (d['ms_per_step'], d['value'], r['kernel'], r['achieved'], r['frac'], r['share_of_step_time']))
x=d['kernel_tflops_extra_steps']
print('   ', {k:v for k,v in x.items() if 'halo3_pc' in k or ('halo3_kernel<2,128' in k)})
PY
  done
done
