# round 5, final evidence part a: GPU suite, default bench line, driver-flag bench line
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r05/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r05/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r05/pytest_gpu.log
SGG_DP_REHEARSE_GLOO=1 python -m pytest tests/test_dp_rccl_gpu.py -x -q > gpurun_out/r05/dp_rccl_test_gloo_rehearsal.log 2>&1 || tail -20 gpurun_out/r05/dp_rccl_test_gloo_rehearsal.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_warmup5.json 2> gpurun_out/r05/bench_steps20_warmup5.err || { tail -20 gpurun_out/r05/bench_steps20_warmup5.err; echo "rc $?"; }
head -c 300 gpurun_out/r05/bench_steps20_warmup5.json; echo
timeout -k 10 500 python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err || { tail -20 gpurun_out/r05/bench.err; echo "rc $?"; }
head -c 300 gpurun_out/r05/bench.json; echo
