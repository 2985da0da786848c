#!/bin/bash
# LN prologue in the forward-only encoder passes (SGG_LN_FUSION=1) against the default, same box, two repetitions
set -e
mkdir -p gpurun_out/lnf
for rep in 1 2; do
  for m in 0 1; do
    SGG_LN_FUSION=$m timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 5 --no-kernel-timing > gpurun_out/lnf/f$m.$rep.json 2> gpurun_out/lnf/f$m.$rep.err
    python -c "import json; d=json.loads(open('gpurun_out/lnf/f$m.$rep.json').read().strip().splitlines()[-1]); print('SGG_LN_FUSION=$m rep $rep: %.2f ms/step  two-stream %.2f' % (d['ms_per_step'], d['two_stream']['ms_per_step']))"
  done
done
