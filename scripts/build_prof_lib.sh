# Instrumented build of the library (cycle accounting inside conv_halo3_kernel): scene-graph-gan_amd/_prof/libsgg_hip_prof.so
set -e
cd "$(dirname "$0")/../scene-graph-gan_amd"
mkdir -p _prof
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -DSGG_HALO_PROFILE -I csrc -c $f -o _prof/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _prof/libsgg_hip_prof.so _prof/*.o
ls -la _prof/libsgg_hip_prof.so
