# same-box A/B of the current library against scene-graph-gan_amd/_prof/libsgg_hip_old.so (scripts/build_old_lib.sh)
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
for rep in 1 2; do for lib in old new; do
  echo -n "$lib: "
  if [ $lib = old ]; then export SGG_HIP_LIB=$GRAFT_REPO_ROOT/scene-graph-gan_amd/_prof/libsgg_hip_old.so; else unset SGG_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-rows 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f triples/s  %.2f ms' % (d['value'], d['ms_per_step']))"
done; done
