# round 5: does the host keep ahead of the GPU in the multi-stream step (no profiler attached)?
set -e
timeout -k 10 300 python scripts/host_lead.py 8 > gpurun_out/r05_host_lead.log 2>&1 || { tail -30 gpurun_out/r05_host_lead.log; exit 1; }
grep -v amdgpu gpurun_out/r05_host_lead.log
