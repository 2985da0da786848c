# round 5: the first k layers of D's next forward beside the generator update's head chain (d_next_layers): tests + A/B
set -e
SGG_OPTIONS="d_next_layers=3" timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_api_gpu.py tests/test_configs34_gpu.py -x -q > gpurun_out/r05_dnl_tests.log 2>&1 || { tail -40 gpurun_out/r05_dnl_tests.log; exit 1; }
tail -2 gpurun_out/r05_dnl_tests.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (d_next_layers off)"
bash scripts/gpu_opt_ab.sh dnl "" "d_next_layers=2" "d_next_layers=3" "d_next_layers=4" "d_next_layers=6" "d_next_layers=3,d_next_cus=0" "d_next_layers=11"
} > gpurun_out/r05_d_next_layers_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_d_next_layers_ab.log
