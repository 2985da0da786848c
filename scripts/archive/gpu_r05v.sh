# round 5: gated kernel timeline (the host has enqueued the whole step before the GPU starts it)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r05_trace_gated
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/trace_step.py 3 > $O/trace_step.log 2> $O/trace.err
cd $GRAFT_REPO_ROOT
rm -rf $O/trace/*.db
python3 scripts/trace_timeline.py $O/trace/t_kernel_trace.csv --gated --gantt > $O/timeline.log
gzip -f $O/trace/t_kernel_trace.csv
cat $O/trace_step.log | grep -v amdgpu; head -40 $O/timeline.log
