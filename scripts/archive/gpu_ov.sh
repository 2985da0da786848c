#!/bin/bash
set -e
mkdir -p gpurun_out/ov
timeout -k 10 500 python -m pytest tests/test_concurrency_gpu.py tests/test_step_gpu.py -x -q > gpurun_out/ov/test.log 2>&1 || { tail -30 gpurun_out/ov/test.log; exit 1; }
tail -2 gpurun_out/ov/test.log
timeout -k 10 400 python scripts/debug_concurrent3.py > gpurun_out/ov/conc3_nopk_all.log 2>&1; grep beside gpurun_out/ov/conc3_nopk_all.log
for rep in 1 2; do
  for m in "" "--overlap-streams"; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 $m > gpurun_out/ov/b$m.$rep.json 2> gpurun_out/ov/b$m.$rep.err
    python -c "import json; d=json.loads(open('gpurun_out/ov/b$m.$rep.json').read().strip().splitlines()[-1]); print('[$m] rep $rep: %.2f ms/step %.1f triples/s  dominant %s avg %.1f us frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel'][:30], 1e3*d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
  done
done
