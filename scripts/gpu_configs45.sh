set -e
cd $GRAFT_REPO_ROOT
echo "config 4 (V=70000)"
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --cpu-rows 0 --vocab 70000 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], d['losses'])"
echo "config 5 (B=32, 448x448)"
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --cpu-rows 0 --batch 32 --size 448 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], d['losses'])"
