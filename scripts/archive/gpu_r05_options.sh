# round 5: the step / schedule / configs parity tests under the non-default values of the run-time options (do the other paths still hold?)
set -e
: > gpurun_out/r05_option_paths_tests.log
for v in "presplit=0" "ln_fusion=0" "ln_fusion=2" "halo_pc64=0" "c3_ln_bwd_fused=0" "g_early=0" "wgrad_late=0" "halo_pc=0" "g_early_cus=0" "presplit_head_grad=0"; do
  echo "## SGG_OPTIONS=$v" >> gpurun_out/r05_option_paths_tests.log
  SGG_OPTIONS="$v" timeout -k 10 600 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -x -q 2>&1 | tail -3 >> gpurun_out/r05_option_paths_tests.log || { echo "FAILED under $v" >> gpurun_out/r05_option_paths_tests.log; tail -30 gpurun_out/r05_option_paths_tests.log; exit 1; }
done
grep -v "^$\|Docs:" gpurun_out/r05_option_paths_tests.log
