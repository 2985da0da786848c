# round 5: LayerNorm backward on chunks of the batch - does the apply pass hit the Infinity Cache?
set -e
timeout -k 10 400 python scripts/ubench/ln_bwd_chunked.py > gpurun_out/r05_ln_bwd_chunked.log 2>&1 || { tail -30 gpurun_out/r05_ln_bwd_chunked.log; exit 1; }
grep -v amdgpu gpurun_out/r05_ln_bwd_chunked.log
