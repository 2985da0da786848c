# round 5: the N = 2 launch path of bench.py (two ranks on one card over gloo: a rehearsal of what the driver starts with --gpus N)
# with the driver's own flags, and as the driver launches it (torch.distributed.run from outside)
set -e
mkdir -p gpurun_out/r05
SGG_DP_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r05/bench_gpus2_selflaunch_gloo_rehearsal.json 2> gpurun_out/r05/bench_gpus2.err || { tail -30 gpurun_out/r05/bench_gpus2.err; exit 1; }
head -c 400 gpurun_out/r05/bench_gpus2_selflaunch_gloo_rehearsal.json; echo
SGG_DP_BACKEND=gloo timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r05/bench_gpus2_torchrun_gloo_rehearsal.json 2> gpurun_out/r05/bench_gpus2_torchrun.err || { tail -30 gpurun_out/r05/bench_gpus2_torchrun.err; exit 1; }
head -c 400 gpurun_out/r05/bench_gpus2_torchrun_gloo_rehearsal.json; echo
