# round 5: the LN-prologue plan re-measured with this round's kernels (LN6 = conv2_5's input; LN0 / LN1 in passes with a backward)
set -e
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the plan in force"
bash scripts/gpu_opt_ab.sh lnplan "" "ln_fusion_skip_bwd=6" "ln_fusion_skip=6" "ln_fusion_skip_bwd=1+6" "ln_fusion_force=4+5"
} > gpurun_out/r05_ln_plan_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_ln_plan_ab.log
