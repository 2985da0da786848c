# GPU evidence of a round, one step per call (gpurun budget): bash scripts/gpu_round.sh <step> [tag]
#   tests   : pytest -m gpu (log under gpurun_out/<tag>/)
#   bench   : the default bench line (+ configs[3], configs[4] lines)
#   stats   : rocprofv3 --kernel-trace --stats of the bench command
#   pmc     : two separate --pmc passes (FETCH_SIZE / WRITE_SIZE) for the roofline `traffic` field
set -e
STEP=${1:-tests}
TAG=${2:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
case $STEP in
  tests)
    timeout -k 10 1100 python -m pytest tests -m gpu -q -x -s > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
    tail -5 $O/pytest_gpu.log ;;
  bench)
    timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
    head -c 400 $O/bench.json ;;
  configs34)
    timeout -k 10 500 python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_config3.err || { tail -20 $O/bench_config3.err; exit 1; }
    timeout -k 10 500 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err || { tail -20 $O/bench_config4.err; exit 1; }
    head -c 300 $O/bench_config3.json; echo; head -c 300 $O/bench_config4.json ;;
  stats)
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/bench.py --steps 5 --warmup 2 --single-stream --serial-steps 0 --other-configs 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 > $O/bench_under_rocprof.json 2> $O/stats.err
    rm -f $O/stats/k_kernel_trace.csv
    head -30 $O/stats/k_kernel_stats.csv ;;
  pmc)
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --single-stream --serial-steps 0 --other-configs 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --no-kernel-timing > $O/pmc_fetch.json 2> $O/pmc_fetch.err
    timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --single-stream --serial-steps 0 --other-configs 0 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --no-kernel-timing > $O/pmc_write.json 2> $O/pmc_write.err
    rm -f $O/pmc_fetch/f_kernel_trace.csv $O/pmc_write/w_kernel_trace.csv
    ls -la $O/pmc_fetch $O/pmc_write ;;
esac
