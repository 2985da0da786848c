# round 5: option d_tail (D's forward of the next critic update behind the generator update's critic head, not its tail): tests + A/B
set -e
python -m pytest tests/test_concurrency_gpu.py tests/test_step_gpu.py tests/test_api_gpu.py -x -q > gpurun_out/r05_dtail_tests.log 2>&1 || { tail -30 gpurun_out/r05_dtail_tests.log; exit 1; }
tail -2 gpurun_out/r05_dtail_tests.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (d_tail=1, g_early_cus=28)"
bash scripts/gpu_opt_ab.sh dtail "" "d_tail=0" "d_tail=0,g_early_cus=0"
} > gpurun_out/r05_d_tail_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_d_tail_ab.log
