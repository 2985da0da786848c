# 4 ranks sharing cuda:0 over gloo (the driver's N=4 launch line with a small workload)
set -e
cd $GRAFT_REPO_ROOT
SGG_DP_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 2 --warmup 1 --batch 4 --size 64 --vocab 50 --cpu-rows 0 2>&1 | tail -2 | cut -c1-400
