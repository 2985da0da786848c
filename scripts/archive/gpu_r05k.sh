# round 5: producer waves of the producer / consumer kernel split by role (weight waves / patch waves): tests, per-shape rates, A/B
set -e
timeout -k 10 900 python -m pytest tests/test_presplit_gpu.py tests/test_kernels_gpu.py -k "presplit or four_block or halo or dma" -x -q > gpurun_out/r05_splitp_tests.log 2>&1 || { tail -40 gpurun_out/r05_splitp_tests.log; exit 1; }
tail -2 gpurun_out/r05_splitp_tests.log
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py tests/test_fullsize_conv_gpu.py -x -q > gpurun_out/r05_splitp_tests2.log 2>&1 || { tail -40 gpurun_out/r05_splitp_tests2.log; exit 1; }
tail -2 gpurun_out/r05_splitp_tests2.log
bash scripts/gpu_r05j.sh > gpurun_out/r05_pc64_per_shape_split.log 2>&1; cat gpurun_out/r05_pc64_per_shape_split.log | cut -c1-180
{
echo "# two-stream schedule, 10 timed steps, interleaved; base = producers split by role + four-block tiles (in tree); nosplitp = -DPC_SPLIT_PRODUCERS=0"
bash scripts/gpu_ab.sh splitp base nosplitp
} > gpurun_out/r05_split_producers_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_split_producers_ab.log
