# A/B on one box: bash scripts/gpu_ab.sh ENVVAR  -> bench with ENVVAR=0 and =1, twice each, interleaved
cd $GRAFT_REPO_ROOT
V=$1
for rep in 1 2; do for val in 0 1; do
  echo -n "$V=$val: "
  env $V=$val timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f triples/s  %.2f ms' % (d['value'], d['ms_per_step']))"
done; done
