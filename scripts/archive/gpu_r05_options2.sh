# round 5: parity tests under the remaining non-default option values
set -e
: > gpurun_out/r05_option_paths_tests2.log
for v in "halo_pc=0" "conv_halo=0" "presplit=0,ln_fusion=2" "ln_fusion_force=4+5+7+8,ln_fusion_force_bwd=4+5" "fwd_cus=24,d_side_cus=24"; do
  echo "## SGG_OPTIONS=$v" >> gpurun_out/r05_option_paths_tests2.log
  SGG_OPTIONS="$v" timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -x -q 2>&1 | tail -3 >> gpurun_out/r05_option_paths_tests2.log || { echo "FAILED under $v" >> gpurun_out/r05_option_paths_tests2.log; tail -30 gpurun_out/r05_option_paths_tests2.log; exit 1; }
done
grep -v "^$\|Docs:" gpurun_out/r05_option_paths_tests2.log
