#!/bin/bash
# kernel trace of the default (two-stream) schedule for scripts/trace_timeline.py:  bash scripts/gpu_trace.sh <tag>
set -e
TAG=${1:-trace}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --serial-steps 0 --other-configs 0 --bitwise-iters 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --no-kernel-timing > $O/trace_bench.json 2> $O/trace.err
python3 $GRAFT_REPO_ROOT/scripts/trace_timeline.py $O/trace/t_kernel_trace.csv 3 | tee $O/timeline.log
