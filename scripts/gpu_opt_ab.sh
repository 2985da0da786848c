#!/bin/bash
# Same-box A/B of run-time options (sgg_amd/lib.py: DEFAULT_OPTIONS) in the full step, interleaved repetitions:
#   bash scripts/gpu_opt_ab.sh <tag> "" "halo_pc=0" "ln_fusion_skip=5+6"      ("" = the defaults)
set -e
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    SGG_OPTIONS="$v" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --serial-steps 0 --other-configs 0 --bitwise-iters 0 --no-kernel-timing > gpurun_out/$TAG/v$i.$rep.json 2> gpurun_out/$TAG/v$i.$rep.err
    python -c "import json; d=json.loads(open('gpurun_out/$TAG/v$i.$rep.json').read().strip().splitlines()[-1]); print('[$v] rep $rep: %.2f ms/step  serial %.2f' % (d['ms_per_step'], d.get('serial', {}).get('ms_per_step', float('nan'))))"
  done
done
