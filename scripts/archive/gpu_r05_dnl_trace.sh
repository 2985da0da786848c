# round 5: gated kernel timeline with d_next_layers=2
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r05_trace_gated_dnl2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SGG_OPTIONS="d_next_layers=2"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/trace_step.py 3 > $O/trace_step.log 2> $O/trace.err
cd $GRAFT_REPO_ROOT
rm -rf $O/trace/*.db
python3 scripts/trace_timeline.py $O/trace/t_kernel_trace.csv --gated --gantt > $O/timeline.log
gzip -f $O/trace/t_kernel_trace.csv
head -5 $O/timeline.log
