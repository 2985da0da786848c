# round 5: the round's evidence, part 1: full GPU suite, bench line with the driver's flags, default bench line
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r05/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r05/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r05/pytest_gpu.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_warmup5.json 2> gpurun_out/r05/bench_steps20_warmup5.err || { tail -20 gpurun_out/r05/bench_steps20_warmup5.err; echo "rc $?"; }
head -c 300 gpurun_out/r05/bench_steps20_warmup5.json; echo
