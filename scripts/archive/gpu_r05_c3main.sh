# round 5: conv1_1's filter gradient on the chain's stream (beside conv1_2's filter gradient) instead of behind it: A/B
set -e
SGG_OPTIONS="c3_wgrad_main=1" timeout -k 10 600 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py -x -q 2>&1 | tail -2
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults"
bash scripts/gpu_opt_ab.sh c3main "" "c3_wgrad_main=1"
} > gpurun_out/r05_c3_wgrad_main_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_c3_wgrad_main_ab.log
