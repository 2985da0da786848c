# round 5: per-shape kernel times of the serial schedule under the plan in force and with LN6 unfused in passes with a backward
# (does conv2_5's filter gradient take the LDS-DMA kernel then, and what do the three launches cost?)
set -e
for v in "" "ln_fusion_skip_bwd=6" "ln_fusion_skip=6"; do
SGG_OPTIONS="$v" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 --bitwise-iters 0 --serial-steps 3 --per-shape 2>/dev/null | V="$v" python -c "
import json,sys,os
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('## [%s] two-stream %.2f ms/step, serial %.2f' % (os.environ['V'], p['ms_per_step'], p['serial']['ms_per_step']))
tot=0
for k,v in sorted(p['per_shape'].items()):
    tot+=v['ms_per_step']
    print('  %-78s %s'%(k,v))
print('  total mfma conv kernels %.3f ms/step' % tot)
"
done > gpurun_out/r05_per_shape_ln6.log 2>&1
tail -5 gpurun_out/r05_per_shape_ln6.log
