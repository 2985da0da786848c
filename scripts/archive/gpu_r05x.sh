# round 5: ln_finalize_kernel at wave priority 3 (in tree) against priority 0: A/B + gated timeline
set -e
{
echo "# two-stream schedule, 10 timed steps, interleaved; base = ln_finalize_kernel at s_setprio 3 (in tree); finprio0 = -DSGG_LN_FIN_PRIO=0"
bash scripts/gpu_ab.sh finprio base finprio0
} > gpurun_out/r05_ln_finalize_prio_ab.log 2>&1
grep -v amdgpu gpurun_out/r05_ln_finalize_prio_ab.log
O=$GRAFT_REPO_ROOT/gpurun_out/r05_trace_gated2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/trace_step.py 3 > $O/trace_step.log 2> $O/trace.err
cd $GRAFT_REPO_ROOT
rm -rf $O/trace/*.db
python3 scripts/trace_timeline.py $O/trace/t_kernel_trace.csv --gated --gantt > $O/timeline.log
gzip -f $O/trace/t_kernel_trace.csv
head -5 $O/timeline.log; grep "ln_finalize_kernel" $O/timeline.log | tail -12
