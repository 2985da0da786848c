#!/bin/bash
# secondary bench lines of a round: the N = 2 launch path with two ranks sharing the card over gloo (rehearsal of the data-parallel
# schedule), and the reference's own image size 221 x 221 (odd sizes run on even canvases)
set -e
TAG=${1:-r04}
O=gpurun_out/$TAG
mkdir -p $O
SGG_DP_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --cpu-rows 0 --other-configs 0 --f32-steps 2 --ci10-steps 0 --serial-steps 2 > $O/bench_gpus2_selflaunch_gloo_rehearsal.json 2> $O/bench_gpus2.err || { tail -30 $O/bench_gpus2.err; exit 1; }
head -c 300 $O/bench_gpus2_selflaunch_gloo_rehearsal.json; echo
timeout -k 10 500 python bench.py --size 221 --cpu-rows 0 --other-configs 0 --f32-steps 3 --ci10-steps 0 > $O/bench_size221.json 2> $O/bench_size221.err || { tail -30 $O/bench_size221.err; exit 1; }
head -c 300 $O/bench_size221.json; echo
