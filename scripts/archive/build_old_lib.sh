# Build the library as of a git revision (default HEAD) into scene-graph-gan_amd/_prof/libsgg_hip_old.so for same-box A/B runs:
#   bash scripts/build_old_lib.sh [rev]; then on the GPU box: SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_old.so python bench.py
set -e
REV=${1:-HEAD}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d)"
mkdir -p "$ROOT/scene-graph-gan_amd/_prof"
git -C "$ROOT" archive "$REV" scene-graph-gan_amd/csrc | tar -x -C "$TMP"
cd "$TMP/scene-graph-gan_amd"
pids=()
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -Xclang -target-feature -Xclang -packed-fp32-ops -I csrc -c "$f" -o "$(basename "$f").o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done      # a failed compile aborts the script here (set -e)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/scene-graph-gan_amd/_prof/libsgg_hip_old.so" *.o
ls -la "$ROOT/scene-graph-gan_amd/_prof/libsgg_hip_old.so"
rm -rf "$TMP"
