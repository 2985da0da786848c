# round 5: per-shape kernel table of configs[4] (batch 32, 448x448) and configs[3] (vocab 70 000): any layer on a fallback kernel?
set -e
for c in 4 3; do
timeout -k 10 400 python bench.py --config $c --steps 3 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 --bitwise-iters 0 --serial-steps 3 --per-shape 2>/dev/null | C=$c python -c "
import json,sys,os
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('## configs[%s]: two-stream %.2f ms/step (%.0f triples/s), serial %.2f' % (os.environ['C'], p['ms_per_step'], p['value'], p['serial']['ms_per_step']))
for k,v in sorted(p['per_shape'].items()):
    print('  %-78s %s'%(k,v))
"
done > gpurun_out/r05_per_shape_configs34.log 2>&1
cat gpurun_out/r05_per_shape_configs34.log | cut -c1-170
