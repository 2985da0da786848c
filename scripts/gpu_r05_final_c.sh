# round 5, final check of the committed tree: smoke, the GPU suite, the two bench lines (now with `traffic` from the calibrated PMC passes)
set -e
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.log 2>&1 || { tail -20 gpurun_out/r05/smoke.log; exit 1; }
tail -1 gpurun_out/r05/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r05/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r05/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r05/pytest_gpu.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_warmup5.json 2> gpurun_out/r05/bench_steps20_warmup5.err || { tail -20 gpurun_out/r05/bench_steps20_warmup5.err; echo "rc $?"; }
head -c 250 gpurun_out/r05/bench_steps20_warmup5.json; echo
timeout -k 10 500 python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err || { tail -20 gpurun_out/r05/bench.err; echo "rc $?"; }
head -c 250 gpurun_out/r05/bench.json; echo
