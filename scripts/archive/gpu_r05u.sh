# round 5: kernel timeline with the head's optimiser step beside the encoder backward
set -e
SGG_OPTIONS="" bash scripts/gpu_trace.sh r05_trace_adamhead > /dev/null
rm -rf gpurun_out/r05_trace_adamhead/trace/*.db
gzip -f gpurun_out/r05_trace_adamhead/trace/t_kernel_trace.csv
head -5 gpurun_out/r05_trace_adamhead/timeline.log
