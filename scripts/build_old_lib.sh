# Build the library as of a git revision (default HEAD) into scene-graph-gan_amd/_prof/libsgg_hip_old.so for same-box A/B runs:
#   bash scripts/build_old_lib.sh [rev]; then on the GPU box: SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_old.so python bench.py
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
rm -rf /tmp/sgg_old && mkdir -p /tmp/sgg_old scene-graph-gan_amd/_prof
git archive $REV scene-graph-gan_amd/csrc | tar -x -C /tmp/sgg_old
cd /tmp/sgg_old/scene-graph-gan_amd
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -I csrc -c $f -o $(basename $f).o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/scene-graph-gan_amd/_prof/libsgg_hip_old.so *.o
ls -la /root/repo/scene-graph-gan_amd/_prof/libsgg_hip_old.so
