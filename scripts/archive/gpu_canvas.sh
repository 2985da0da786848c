#!/bin/bash
# odd-size canvas path: LN valid-region kernels, 221x221 end-to-end parity, bench at 221 vs 224
set -e
mkdir -p gpurun_out/canvas
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" > gpurun_out/canvas/ln.log 2>&1
timeout -k 10 500 python -m pytest tests/test_configs34_gpu.py -x -q -s -k "realdata or configs1_224" > gpurun_out/canvas/e2e.log 2>&1
timeout -k 10 300 python bench.py --size 221 --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 > gpurun_out/canvas/bench221.json 2> gpurun_out/canvas/bench221.err
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 > gpurun_out/canvas/bench224.json 2> gpurun_out/canvas/bench224.err
tail -3 gpurun_out/canvas/ln.log gpurun_out/canvas/e2e.log
cat gpurun_out/canvas/bench221.json gpurun_out/canvas/bench224.json | cut -c1-400
