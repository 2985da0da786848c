cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for halo in 1 0; do
  SGG_CONV_HALO=$halo timeout -k 10 120 python scripts/prof_conv.py 64 112 128 128 3 1 6000 fwd_ws > gpurun_out/pw$halo.log 2>&1 &
  PID=$!
  sleep 14
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|power\|Socket" | head -6; sleep 0.5; done
  wait $PID
  cat gpurun_out/pw$halo.log | tail -1
done
