set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], d['roofline']['traffic'])"
