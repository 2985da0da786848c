# Variant of the library that differs from the in-tree build in ONE source file: bash scripts/build_variant_one.sh <name> <file.hip> -D...
#   -> scene-graph-gan_amd/_prof/libsgg_hip_<name>.so (the other objects come from scene-graph-gan_amd/_build, i.e. build.py's last build)
set -e
VARIANT=$1; SRC=$2; shift; shift
cd "$(dirname "$0")/../scene-graph-gan_amd"
mkdir -p _prof/$VARIANT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -I csrc -c csrc/$SRC -o _prof/$VARIANT/$SRC.o 2> >(grep -v "not a recognized feature" >&2)
OBJS=$(ls _build/*.hip.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _prof/libsgg_hip_$VARIANT.so _prof/$VARIANT/$SRC.o $OBJS
ls -la _prof/libsgg_hip_$VARIANT.so
