# train.py soak on synthetic data (defaults: two streams, G's encoder once per iteration): 150 iterations x (5 critic + 1 generator) updates,
# then resume from the checkpoint for 50 more; every logged loss must be finite
set -e
mkdir -p gpurun_out/soak && rm -rf /tmp/sgg_soak_ck gpurun_out/soak/logs
A="--synthetic 64,224,1000 --critic_iters 5 --checkpoints_dir /tmp/sgg_soak_ck --summaries_dir gpurun_out/soak/logs"
timeout -k 10 500 python train.py $A --max_iterations 150 > gpurun_out/soak/run1.log 2>&1 || { tail -20 gpurun_out/soak/run1.log; exit 1; }
timeout -k 10 300 python train.py $A --max_iterations 200 --resume True > gpurun_out/soak/run2.log 2>&1 || { tail -20 gpurun_out/soak/run2.log; exit 1; }
python - <<'PY'
import json, math
recs = [json.loads(l) for l in open("gpurun_out/soak/logs/losses.jsonl")]
assert recs and all(math.isfinite(r[k]) for r in recs for k in ("disc_loss", "gen_loss", "gp")), recs[-3:]
print(len(recs), "records; first", recs[0], "last", recs[-1])
PY
