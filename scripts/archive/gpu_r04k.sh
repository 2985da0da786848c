#!/bin/bash
# round 4, call k: the head's gradient converted to the pre-split format (sgg_presplit16): GPU suite, same-box A/B against the
# option presplit_head_grad=0, per-call rates of the serial steps
set -e
mkdir -p gpurun_out/r04k
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04k/pytest.log 2>&1 || { tail -40 gpurun_out/r04k/pytest.log; exit 1; }
tail -2 gpurun_out/r04k/pytest.log
bash scripts/gpu_opt_ab.sh r04k_opt "" "presplit_head_grad=0"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 --per-shape > gpurun_out/r04k/serial.json 2> gpurun_out/r04k/serial.err
