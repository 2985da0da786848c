# quick per-kernel stats of the default bench (rocprofv3 --kernel-trace --stats): bash scripts/gpu_kstats.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kstats
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 > $O/bench.json 2> $O/err.log
rm -f $O/k_kernel_trace.csv
