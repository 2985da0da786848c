# round 5: conv1_1's filter gradient fused with the LayerNorm backward's apply pass: tests + full-step A/B
set -e
python -m pytest tests/test_kernels_gpu.py -k "fused_with_layernorm or conv_fwd_dgrad_wgrad or deferred_finalize" -x -q > gpurun_out/r05_c3ln_tests.log 2>&1 || { tail -40 gpurun_out/r05_c3ln_tests.log; exit 1; }
tail -2 gpurun_out/r05_c3ln_tests.log
python -m pytest tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_configs34_gpu.py -x -q > gpurun_out/r05_c3ln_tests2.log 2>&1 || { tail -40 gpurun_out/r05_c3ln_tests2.log; exit 1; }
tail -2 gpurun_out/r05_c3ln_tests2.log
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (c3_ln_bwd_fused=1)"
bash scripts/gpu_opt_ab.sh c3ln "" "c3_ln_bwd_fused=0"
} > gpurun_out/r05_c3_ln_bwd_fused_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_c3_ln_bwd_fused_ab.log
