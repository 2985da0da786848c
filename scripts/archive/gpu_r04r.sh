#!/bin/bash
# round 4, call r: conv1_1 kernels with LDS-only barriers (no vmcnt(0) drain per tile) against __syncthreads()
set -e
mkdir -p gpurun_out/r04r
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -x > gpurun_out/r04r/pytest.log 2>&1 || { tail -40 gpurun_out/r04r/pytest.log; exit 1; }
tail -2 gpurun_out/r04r/pytest.log
bash scripts/gpu_ab.sh r04r_ab base c3sync
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 > gpurun_out/r04r/serial.json 2> gpurun_out/r04r/serial.err
python -c "
import json; d=json.loads(open('gpurun_out/r04r/serial.json').read().strip().splitlines()[-1])
print([ (r['kernel'], round(r['avg_us'],1), round(r['frac_of_8TBps'],3)) for r in d['roofline_hbm'] if 'c3' in r['kernel']])"
