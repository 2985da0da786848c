# Rehearse the N>1 launch path of bench.py on a 1-GPU box: 2 ranks share cuda:0, gradients all-reduced over gloo.
set -e
cd $GRAFT_REPO_ROOT
SGG_DP_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --size 64 --vocab 50 --cpu-rows 0 2>&1 | tail -3
timeout -k 10 300 python bench.py --gpus 1 --steps 2 --warmup 1 --batch 16 --size 64 --vocab 50 --cpu-rows 0 2>&1 | tail -1
