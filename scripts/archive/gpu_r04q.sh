#!/bin/bash
# round 4, call q: conv1_1 kernels on f32 MFMA: kernel tests, per-kernel timing, full-step A/B against the previous library; main
# stream priority
set -e
mkdir -p gpurun_out/r04q
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_configs34_gpu.py tests/test_fullsize_conv_gpu.py -m gpu -q -x > gpurun_out/r04q/pytest.log 2>&1 || { tail -40 gpurun_out/r04q/pytest.log; exit 1; }
tail -2 gpurun_out/r04q/pytest.log
bash scripts/gpu_ab.sh r04q_ab base oldc3
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 > gpurun_out/r04q/serial.json 2> gpurun_out/r04q/serial.err
python -c "
import json; d=json.loads(open('gpurun_out/r04q/serial.json').read().strip().splitlines()[-1])
print([ (r['kernel'], round(r['avg_us'],1), round(r['frac_of_8TBps'],3)) for r in d['roofline_hbm'] if 'c3' in r['kernel']])"
