# Round-end GPU evidence: full gpu test suite, default bench line, rocprofv3 kernel stats of the same command, and
# two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the roofline `traffic` field. Usage: bash scripts/gpu_profile_round.sh r01
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 > $O/bench_under_rocprof.json 2> $O/stats.err
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-rows 0 --no-kernel-timing > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-rows 0 --no-kernel-timing > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
timeout -k 10 300 python bench.py --conv-precision 0 --cpu-rows 0 > $O/bench_precision0_native_f32_mfma.json 2>/dev/null
timeout -k 10 300 python bench.py --conv-precision 6 --cpu-rows 0 > $O/bench_precision6_bf16x6.json 2>/dev/null
timeout -k 10 300 python bench.py --conv-precision 3 --cpu-rows 0 > $O/bench_precision3_bf16x3.json 2>/dev/null
timeout -k 10 300 python bench.py --critic-iters 10 --steps 2 --warmup 1 --cpu-rows 0 > $O/bench_critic_iters10.json 2>/dev/null
bash scripts/gpu_configs45.sh > $O/configs45.log 2>&1
bash scripts/gpu_dp_rehearsal.sh > $O/dp_rehearsal.log 2>&1
rm -f $O/stats/k_kernel_trace.csv     # large; the per-kernel stats summary is what gets committed
ls -R $O | head -40
