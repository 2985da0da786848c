# Instrumented / experimental builds of the library into scene-graph-gan_amd/_prof/ (selected at run time with SGG_HIP_LIB=...):
#   bash scripts/build_prof_lib.sh            -DSGG_HALO_PROFILE  -> libsgg_hip_prof.so   (cycle accounting inside conv_halo3_kernel)
#   bash scripts/build_prof_lib.sh nosplit    -DSGG_EXPERIMENT_NOSPLIT -> libsgg_hip_nosplit.so (timing only: no residual fp16 piece)
set -e
VARIANT=${1:-prof}
DEF=-DSGG_HALO_PROFILE
[ "$VARIANT" = nosplit ] && DEF=-DSGG_EXPERIMENT_NOSPLIT
cd "$(dirname "$0")/../scene-graph-gan_amd"
mkdir -p _prof/$VARIANT
pids=()
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -Xclang -target-feature -Xclang -packed-fp32-ops $DEF -I csrc -c $f -o _prof/$VARIANT/$(basename $f).o  2> >(grep -v "not a recognized feature" >&2) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _prof/libsgg_hip_$VARIANT.so _prof/$VARIANT/*.o
ls -la _prof/libsgg_hip_$VARIANT.so
