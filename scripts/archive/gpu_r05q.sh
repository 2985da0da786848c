# round 5: kernel timelines of the multi-stream step under the plan in force and with LN6 unfused in passes with a backward
set -e
SGG_OPTIONS="" bash scripts/gpu_trace.sh r05_trace_default > /dev/null
SGG_OPTIONS="ln_fusion_skip_bwd=6" bash scripts/gpu_trace.sh r05_trace_skipbwd6 > /dev/null
rm -rf gpurun_out/r05_trace_default/trace/*.db gpurun_out/r05_trace_skipbwd6/trace/*.db
gzip -f gpurun_out/r05_trace_default/trace/t_kernel_trace.csv gpurun_out/r05_trace_skipbwd6/trace/t_kernel_trace.csv
ls -la gpurun_out/r05_trace_default/trace gpurun_out/r05_trace_skipbwd6/trace
tail -3 gpurun_out/r05_trace_default/timeline.log gpurun_out/r05_trace_skipbwd6/timeline.log
