# round 5: per-launch CU cap of the forwards that run beside a latency-critical chain (options g_early_cus, d_side_cus): full-step A/B
set -e
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (all 32 CUs per XCD)"
bash scripts/gpu_opt_ab.sh cuopt "" "g_early_cus=28" "g_early_cus=24" "d_side_cus=28" "g_early_cus=28,d_side_cus=28" "g_early_cus=20"
} > gpurun_out/r05_early_forward_cu_cap_ab.log 2>&1
cat gpurun_out/r05_early_forward_cu_cap_ab.log
