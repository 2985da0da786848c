# G-encoder reuse inside one iteration (train.py default): parity test, then train.py timing with and without it on the same box
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "reuse" > gpurun_out/reuse_test.log 2>&1 || { tail -30 gpurun_out/reuse_test.log; exit 1; }
tail -3 gpurun_out/reuse_test.log
for ci in 10 5; do
timeout -k 10 300 python train.py --synthetic 64,224,1000 --critic_iters $ci --max_iterations 40 > gpurun_out/reuse_on_ci$ci.log 2>&1
timeout -k 10 300 python train.py --synthetic 64,224,1000 --critic_iters $ci --max_iterations 40 --recompute_generator_encoder > gpurun_out/reuse_off_ci$ci.log 2>&1
done
tail -n 2 gpurun_out/reuse_on_ci10.log gpurun_out/reuse_off_ci10.log gpurun_out/reuse_on_ci5.log gpurun_out/reuse_off_ci5.log
