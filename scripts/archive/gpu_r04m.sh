#!/bin/bash
# round 4, call m: filter gradient j started behind dgrad_j (option wgrad_late) against beside it; two-stream schedule
set -e
mkdir -p gpurun_out/r04m
SGG_OPTIONS="wgrad_late=1" timeout -k 10 600 python -m pytest tests/test_concurrency_gpu.py -m gpu -q -x > gpurun_out/r04m/pytest.log 2>&1 || { tail -40 gpurun_out/r04m/pytest.log; exit 1; }
tail -2 gpurun_out/r04m/pytest.log
bash scripts/gpu_opt_ab.sh r04m_opt "" "wgrad_late=1"
# kernel trace of the two-stream schedule (timeline analysis: scripts/trace_timeline.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r04m
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --serial-steps 0 --other-configs 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --no-kernel-timing > $O/trace_bench.json 2> $O/trace.err
ls -la $O/trace
