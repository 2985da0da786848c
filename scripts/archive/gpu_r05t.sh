# round 5: the head's optimiser step beside the encoder backward (adam_head_early): tests + A/B
set -e
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py tests/test_api_gpu.py -x -q > gpurun_out/r05_adam_head_tests.log 2>&1 || { tail -40 gpurun_out/r05_adam_head_tests.log; exit 1; }
tail -2 gpurun_out/r05_adam_head_tests.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (adam_head_early on)"
bash scripts/gpu_opt_ab.sh adamhead "" "adam_head_early=0"
} > gpurun_out/r05_adam_head_early_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_adam_head_early_ab.log
