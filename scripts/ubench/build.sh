# Build the micro-experiments (binaries are not tracked): bash scripts/ubench/build.sh
set -e
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o "${f%.hip}.bin" "$f"
done
ls -la *.bin
