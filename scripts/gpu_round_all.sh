# the round's bench evidence in two calls (gpurun limit 1200 s each): bash scripts/gpu_round_all.sh {a|b} <tag>
#   a: default bench line, the driver-flag line (--steps 20 --warmup 5), rocprofv3 kernel statistics
#   b: the two PMC passes, configs[3] and configs[4] lines
set -e
PART=${1:-a}
TAG=${2:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$PART" = a ]; then
  bash scripts/gpu_round.sh bench $TAG && echo
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err || { tail -20 $O/bench_steps20_warmup5.err; exit 1; }
  head -c 300 $O/bench_steps20_warmup5.json && echo
  bash scripts/gpu_round.sh stats $TAG
else
  bash scripts/gpu_round.sh pmc $TAG
  bash scripts/gpu_round.sh configs34 $TAG
fi
