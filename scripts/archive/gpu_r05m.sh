# round 5: LN prologue forced in forward-only passes for the consumers on the producer / consumer kernel (LN4, LN5, LN7, LN8)
set -e
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the plan in force"
bash scripts/gpu_opt_ab.sh lnplan2 "" "ln_fusion_force=4+5" "ln_fusion_force=4+5+7+8" "ln_fusion_force=5" "ln_fusion_force=4" "ln_fusion_force=7+8"
} > gpurun_out/r05_ln_plan_ab2.log 2>&1
grep -v amdgpu gpurun_out/r05_ln_plan_ab2.log
