# the default bench line (full: roofline, parity incl. B = 64 gradients, cpu_baseline) + the same timed region on a variant library
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r03b}
mkdir -p $O
timeout -k 10 700 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench rc $?"; tail -20 $O/bench.err; }
head -c 600 $O/bench.json; echo
for v in ${VARIANTS:-nopc}; do
  SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so timeout -k 10 300 python bench.py --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --two-stream-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err || { echo "bench $v rc $?"; tail -20 $O/bench_$v.err; }
  head -c 300 $O/bench_$v.json; echo
done
