# four-wave 3x3 kernel on the K = 32 MFMA shape for the LN-prologue forwards (csrc/conv_halo_k32.hip): kernel tests (layout 4 + prologue
# now reach it), per-kernel timing against the four-wave 32x32x16 kernel, then the whole step with the prologue layers on layout 4
# (SGG_HALO_PC_LNP=1 -> K = 32 kernel) against layout 1 (conv_halo3_kernel)
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-k32}
mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_kernels_gpu.py -k "halo or prologue or epilogue or prepare_weights" -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
SGG_HALO_PC_LNP=1 timeout -k 10 500 python -m pytest tests/test_fullsize_conv_gpu.py tests/test_step_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x -k "not routing" > $O/t3.log 2>&1 || { tail -40 $O/t3.log; exit 1; }
tail -2 $O/t3.log
for rep in 1 2; do
  for v in lay1 k32; do
    if [ "$v" = lay1 ]; then unset SGG_HALO_PC_LNP; else export SGG_HALO_PC_LNP=1; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 0 > $O/$v.$rep.json 2> $O/$v.$rep.err
    python - <<PY | tee -a $O/ab.log
import json
d=json.loads(open('$O/$v.$rep.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$v rep $rep: %.2f ms/step  %.1f triples/s  dominant %s %.1f TF frac %.3f share %.3f' % (d['ms_per_step'], d['value'], r['kernel'], r['achieved'], r['frac'], r['share_of_step_time']))
x=d['kernel_tflops_extra_steps']
print('   ', {k:round(v,1) for k,v in x.items() if 'halo3_pc' in k or 'k32' in k or ('halo3_kernel<2,128' in k)})
PY
  done
done
