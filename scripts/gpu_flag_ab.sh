#!/bin/bash
# Same-box A/B of bench.py FLAGS in the full step, interleaved repetitions:  bash scripts/gpu_flag_ab.sh <tag> "" "--head-side-stream"
set -e
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --serial-steps 0 --other-configs 0 --bitwise-iters 0 --no-kernel-timing $v > gpurun_out/$TAG/v$i.$rep.json 2> gpurun_out/$TAG/v$i.$rep.err
    python -c "import json; d=json.loads(open('gpurun_out/$TAG/v$i.$rep.json').read().strip().splitlines()[-1]); print('[$v] rep $rep: %.2f ms/step' % d['ms_per_step'])"
  done
done
