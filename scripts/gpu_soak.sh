# short soak: the drop-in train.py on synthetic data at the headline config, 40 iterations; prints the loss log tail
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/sgg_soak
timeout -k 10 500 python train.py --synthetic 64,224,1000 --critic_iters 1 --max_iterations 40 --checkpoints_dir /tmp/sgg_soak/ckpt --summaries_dir /tmp/sgg_soak/sum 2>&1 | tail -12
