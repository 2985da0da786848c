#!/bin/bash
# round 4, call e: LN-prologue plan A/Bs on top of the pre-split activations
set -e
bash scripts/gpu_opt_ab.sh r04e_ab "" "ln_fusion=0" "ln_fusion_skip=4+5" "ln_fusion_skip=4+5+7+8" "ln_fusion_skip=4+5+6+7+8+9"
