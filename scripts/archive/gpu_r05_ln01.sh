# round 5: LN0 / LN1 unfused in passes with a backward (conv1_2's / conv1_3's filter gradients - the last matrix kernels of every
# backward - then stage a pre-split activation instead of normalising y on the fly): A/B
set -e
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; [] = the plan in force"
bash scripts/gpu_opt_ab.sh ln01 "" "ln_fusion_skip_bwd=0" "ln_fusion_skip_bwd=1" "ln_fusion_skip_bwd=0+1"
} > gpurun_out/r05_ln_plan_ab4.log 2>&1
grep -v amdgpu gpurun_out/r05_ln_plan_ab4.log
