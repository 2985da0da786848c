#!/bin/bash
# round 4, call v: g_early as default: GPU suite, default bench line
set -e
bash scripts/gpu_round.sh tests r04
bash scripts/gpu_round.sh bench r04
