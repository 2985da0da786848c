# round 5: D's forward of the next critic update beside the generator update's tail (d_next_gate): tests + A/B
set -e
SGG_OPTIONS="d_next_gate=0" timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_api_gpu.py -x -q > gpurun_out/r05_d_next_tests.log 2>&1 || { tail -40 gpurun_out/r05_d_next_tests.log; exit 1; }
tail -2 gpurun_out/r05_d_next_tests.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (d_next_gate off)"
bash scripts/gpu_opt_ab.sh dnext "" "d_next_gate=0" "d_next_gate=1" "d_next_gate=3" "d_next_gate=0,d_next_cus=28" "d_next_gate=6"
} > gpurun_out/r05_d_next_gate_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_d_next_gate_ab.log
