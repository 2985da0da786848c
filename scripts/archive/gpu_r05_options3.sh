# round 5: parity tests with the resident convolution kernels off (every layer on the gather kernels)
: > gpurun_out/r05_option_paths_tests3.log
for v in "conv_halo=0"; do
  echo "## SGG_OPTIONS=$v" >> gpurun_out/r05_option_paths_tests3.log
  SGG_OPTIONS="$v" timeout -k 10 1000 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -q 2>&1 | tail -25 >> gpurun_out/r05_option_paths_tests3.log
done
grep -v "^$\|Docs:" gpurun_out/r05_option_paths_tests3.log | cut -c1-250
