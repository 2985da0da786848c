# round 5: Adam step of everything but conv1_1 + weight preparation beside conv1_1's filter gradient (split_adam_tail): tests + A/B
set -e
python -m pytest tests/test_concurrency_gpu.py tests/test_step_gpu.py tests/test_api_gpu.py tests/test_bench_contract.py -m gpu -x -q > gpurun_out/r05_split_tail_tests.log 2>&1 || { tail -40 gpurun_out/r05_split_tail_tests.log; exit 1; }
tail -2 gpurun_out/r05_split_tail_tests.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (split_adam_tail=1)"
bash scripts/gpu_opt_ab.sh splittail "" "split_adam_tail=0"
bash scripts/gpu_opt_ab.sh splittail2 "" "split_adam_tail=0"
} > gpurun_out/r05_split_adam_tail_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_split_adam_tail_ab.log
