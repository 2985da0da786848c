# round 5: LN4 / LN5 fused in forward-only passes by rule (trunk.pc_ln_fusion_pays): tests + A/B against skipping them
set -e
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py tests/test_api_gpu.py -x -q > gpurun_out/r05_lnplan_tests.log 2>&1 || { tail -40 gpurun_out/r05_lnplan_tests.log; exit 1; }
tail -2 gpurun_out/r05_lnplan_tests.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the plan in force (LN4, LN5 fused in forward-only passes)"
bash scripts/gpu_opt_ab.sh lnplan3 "" "ln_fusion_skip=4+5"
} > gpurun_out/r05_ln_plan_ab3.log 2>&1
grep -v amdgpu gpurun_out/r05_ln_plan_ab3.log
