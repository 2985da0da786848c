#!/bin/bash
# round 4, call o: cross-step overlap in one direction only
set -e
bash scripts/gpu_opt_ab.sh r04o_opt "" "cross_step=1" "cross_step=2" "cross_step=3"
