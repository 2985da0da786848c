# Experimental build of the library with extra -D flags into scene-graph-gan_amd/_prof/ (selected at run time with SGG_HIP_LIB=...):
#   bash scripts/build_variant_lib.sh <name> -DSGG_LN_NT=1 ...   -> scene-graph-gan_amd/_prof/libsgg_hip_<name>.so
set -e
VARIANT=$1; shift
cd "$(dirname "$0")/../scene-graph-gan_amd"
mkdir -p _prof/$VARIANT
pids=()
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -I csrc -c $f -o _prof/$VARIANT/$(basename $f).o  2> >(grep -v "not a recognized feature" >&2) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _prof/libsgg_hip_$VARIANT.so _prof/$VARIANT/*.o
ls -la _prof/libsgg_hip_$VARIANT.so
