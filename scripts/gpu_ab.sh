#!/bin/bash
# A/B of library builds in the full step: bash scripts/gpu_ab.sh <tag> <variant>...   ("base" = the in-tree library); two repetitions each
set -e
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --serial-steps 0 --other-configs 0 --bitwise-iters 0 --no-kernel-timing > gpurun_out/$TAG/$v.$rep.json 2> gpurun_out/$TAG/$v.$rep.err
    python -c "import json,sys; d=json.loads(open('gpurun_out/$TAG/$v.$rep.json').read().strip().splitlines()[-1]); print('$v rep $rep: %.2f ms/step  %.1f triples/s' % (d['ms_per_step'], d['value']))"
  done
done
