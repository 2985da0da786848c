set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step']); print(d['roofline']); print(d['kernel_time_s']); print(d['kernel_tflops'])"
