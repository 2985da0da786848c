# round 5, final evidence on ONE box (boxes of the pool differ by several per cent: the rocprofv3 averages and the bench line's own
# HIP-event figures agree only when they come from the same box): kernel statistics, the two PMC passes, then the two bench lines
set -e
mkdir -p gpurun_out/r05
bash scripts/gpu_round.sh stats r05 > /dev/null
bash scripts/gpu_round.sh pmc r05 > /dev/null
python scripts/summarize_profiles.py gpurun_out/r05 r05tmp > /dev/null 2>&1 || true
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_warmup5.json 2> gpurun_out/r05/bench_steps20_warmup5.err || { tail -20 gpurun_out/r05/bench_steps20_warmup5.err; echo "rc $?"; }
head -c 250 gpurun_out/r05/bench_steps20_warmup5.json; echo
timeout -k 10 500 python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err || { tail -20 gpurun_out/r05/bench.err; echo "rc $?"; }
head -c 250 gpurun_out/r05/bench.json; echo
head -4 gpurun_out/r05/stats/k_kernel_stats.csv | cut -c1-140
