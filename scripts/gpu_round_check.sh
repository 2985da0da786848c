set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r01_pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 > $R/gpurun_out/r01_prof_bench.json 2> $R/gpurun_out/r01_prof.err
ls -R $R/gpurun_out/prof_stats | head -30
