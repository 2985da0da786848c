cd $GRAFT_REPO_ROOT
for rep in 1 2; do for flag in "" "--overlap-streams"; do
  echo -n "flag '$flag': "
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-kernel-timing $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f triples/s  %.2f ms' % (d['value'], d['ms_per_step']))"
done; done
