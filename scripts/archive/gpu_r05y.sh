# round 5: 16-byte fill, four tiles per workgroup in the weight transposes: tests + gated timeline + bench
set -e
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_concurrency_gpu.py tests/test_presplit_gpu.py -x -q > gpurun_out/r05_tail_tests.log 2>&1 || { tail -40 gpurun_out/r05_tail_tests.log; exit 1; }
tail -2 gpurun_out/r05_tail_tests.log
O=$GRAFT_REPO_ROOT/gpurun_out/r05_trace_gated3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/trace_step.py 3 > $O/trace_step.log 2> $O/trace.err
cd $GRAFT_REPO_ROOT
rm -rf $O/trace/*.db
python3 scripts/trace_timeline.py $O/trace/t_kernel_trace.csv --gated --gantt > $O/timeline.log
gzip -f $O/trace/t_kernel_trace.csv
head -5 $O/timeline.log; grep "wp_\|fill_kernel\|fill4" $O/timeline.log | tail -12
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --serial-steps 0 --other-configs 0 --bitwise-iters 0 --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: %.2f ms/step' % d['ms_per_step'])"
