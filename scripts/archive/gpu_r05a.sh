# round 5, call 3: counter calibration (with durations), conv1_1 forward with plain / nontemporal stores, full-step A/B of the store policy
set -e
bash scripts/gpu_calib.sh > gpurun_out/calib.out 2>&1 || tail -5 gpurun_out/calib.out
{
echo "# conv1_1 forward (batch 64, 224x224, 3 -> 32, LayerNorm partials in the epilogue), 20 launches back to back, ms per launch"
for v in base c3nt; do
  if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
  for rep in 1 2; do echo -n "$v fwd_stats: "; python scripts/prof_conv.py 64 224 3 32 3 1 20 fwd_stats | tail -1; done
  for rep in 1 2; do echo -n "$v fwd: "; python scripts/prof_conv.py 64 224 3 32 3 1 20 fwd | tail -1; done
done
unset SGG_HIP_LIB
echo "# full step, two-stream schedule: base = conv1_1 plain stores (in tree), c3nt = conv1_1 nontemporal (round 4), ntoff = every convolution epilogue plain"
bash scripts/gpu_ab.sh ntab base c3nt ntoff
} > gpurun_out/r05_store_policy_ab.log 2>&1
tail -12 gpurun_out/r05_store_policy_ab.log
