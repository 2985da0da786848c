# round 5: producers of the register-staged / LN-prologue variants of the producer / consumer kernel split by role: tests + A/B
set -e
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_presplit_gpu.py -k "halo or prologue or presplit or four_block" -x -q > gpurun_out/r05_splitreg_tests.log 2>&1 || { tail -40 gpurun_out/r05_splitreg_tests.log; exit 1; }
tail -2 gpurun_out/r05_splitreg_tests.log
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -x -q > gpurun_out/r05_splitreg_tests2.log 2>&1 || { tail -40 gpurun_out/r05_splitreg_tests2.log; exit 1; }
tail -2 gpurun_out/r05_splitreg_tests2.log
{
echo "# two-stream schedule, 10 timed steps, interleaved; base = register-staged / LN-prologue producers split by role (in tree); nosplitreg = -DPC_SPLIT_PRODUCERS_REG=0"
bash scripts/gpu_ab.sh splitreg base nosplitreg
} > gpurun_out/r05_split_producers_reg_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_split_producers_reg_ab.log
SGG_OPTIONS="" timeout -k 10 300 python bench.py --steps 3 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 --bitwise-iters 0 --serial-steps 3 --per-shape 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in p['per_shape'].items():
    if 'pc_kernel' in k: print('  %-70s %s'%(k,v))
" > gpurun_out/r05_pc_per_shape_lnp.log 2>&1
cat gpurun_out/r05_pc_per_shape_lnp.log
