# LN prologue: per-kernel cost against the plain kernels (same box): bash scripts/gpu_lnp.sh
set -e
cd $GRAFT_REPO_ROOT
for shape in "64 112 128 128" "64 56 256 256" "64 224 32 32"; do
  for mode in fwd_ws fwd_ws_ln wgrad wgrad_ln; do
    timeout -k 10 120 python scripts/prof_conv.py $shape 3 1 10 $mode 2>&1 | grep -v amdgpu.ids
  done
done
