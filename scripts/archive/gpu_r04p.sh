#!/bin/bash
# round 4, call p: priority of the side streams
set -e
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
bash scripts/gpu_opt_ab.sh r04p_opt "" "side_priority=1" "side_priority=-1"
