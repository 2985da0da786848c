# round 5: finer sweep of g_early_cus and the static-halves idea (fwd_cus=16): full-step A/B
set -e
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the defaults (all 32 CUs per XCD)"
bash scripts/gpu_opt_ab.sh cuopt2 "" "g_early_cus=30" "g_early_cus=28" "g_early_cus=26" "fwd_cus=16" "fwd_cus=28"
} > gpurun_out/r05_early_forward_cu_cap_ab2.log 2>&1
cat gpurun_out/r05_early_forward_cu_cap_ab2.log
