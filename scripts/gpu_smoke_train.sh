#!/bin/bash
# the driver's smoke() and a short synthetic training run (the product's default schedule), both through the in-tree library
set -e
mkdir -p gpurun_out/smoke
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 600 python train.py --synthetic 64,224,1000 --critic_iters 1 --max_iterations 12 --checkpoints_dir gpurun_out/smoke/ck --summaries_dir gpurun_out/smoke/sum > gpurun_out/smoke/train.log 2>&1 || { tail -30 gpurun_out/smoke/train.log; exit 1; }
tail -6 gpurun_out/smoke/train.log
timeout -k 10 600 python train.py --synthetic 64,224,1000 --critic_iters 5 --max_iterations 6 --checkpoints_dir gpurun_out/smoke/ck5 --summaries_dir gpurun_out/smoke/sum5 > gpurun_out/smoke/train5.log 2>&1 || { tail -30 gpurun_out/smoke/train5.log; exit 1; }
tail -4 gpurun_out/smoke/train5.log
