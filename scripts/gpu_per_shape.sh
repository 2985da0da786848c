cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-rows 0 --per-shape 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k,v in d['per_shape'].items(): print('%-90s %3d %8.3f ms/step %7.1f TF' % (k, v['launches'], v['ms_per_step'], v['tflops']))"
