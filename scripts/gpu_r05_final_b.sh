# round 5, final evidence part b: rocprofv3 kernel statistics (serial schedule), the two PMC passes, the kernel trace of the multi-stream step
set -e
bash scripts/gpu_round.sh stats r05
bash scripts/gpu_round.sh pmc r05
bash scripts/gpu_trace.sh r05trace2 > /dev/null 2>&1 || true
tail -25 gpurun_out/r05trace2/timeline.log | cut -c1-200
