# rocprofv3 kernel statistics of a short free-running bench (no parity / CPU legs): gpurun_out/<name>/stats.csv
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-stats}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 0 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -1 $O/bench.json | cut -c1-300
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/stats.csv
head -25 $O/stats.csv | cut -c1-200
