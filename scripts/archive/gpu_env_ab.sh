#!/bin/bash
# A/B of an environment switch of sgg_amd/lib.py in the full step: bash scripts/gpu_env_ab.sh <tag> <VAR> <value>...   (two repetitions)
set -e
TAG=$1; VAR=$2; shift; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 5 --no-kernel-timing > gpurun_out/$TAG/$v.$rep.json 2> gpurun_out/$TAG/$v.$rep.err
    python -c "import json; d=json.loads(open('gpurun_out/$TAG/$v.$rep.json').read().strip().splitlines()[-1]); print('$VAR=$v rep $rep: %.2f ms/step  two-stream %.2f' % (d['ms_per_step'], d['two_stream']['ms_per_step']))"
  done
done
