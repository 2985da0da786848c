# round 5: four-block (64-column) tiles of the producer / consumer kernel for the dgrads of conv2_2 / conv2_3: tests + A/B
set -e
timeout -k 10 600 python -m pytest tests/test_presplit_gpu.py -k "four_block" -x -q > gpurun_out/r05_pc64_tests.log 2>&1 || { tail -40 gpurun_out/r05_pc64_tests.log; exit 1; }
tail -2 gpurun_out/r05_pc64_tests.log
timeout -k 10 900 python -m pytest tests/test_presplit_gpu.py tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -x -q > gpurun_out/r05_pc64_tests2.log 2>&1 || { tail -40 gpurun_out/r05_pc64_tests2.log; exit 1; }
tail -2 gpurun_out/r05_pc64_tests2.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (halo_pc64=1)"
bash scripts/gpu_opt_ab.sh pc64 "" "halo_pc64=0"
bash scripts/gpu_opt_ab.sh pc64b "" "halo_pc64=0"
} > gpurun_out/r05_pc64_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_pc64_ab.log
