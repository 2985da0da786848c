#!/bin/bash
set -e
mkdir -p gpurun_out/ln
timeout -k 10 200 python scripts/prof_ln.py > gpurun_out/ln/base.log 2>&1
for v in "$@"; do
  SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$v.so timeout -k 10 200 python scripts/prof_ln.py > gpurun_out/ln/$v.log 2>&1
done
for f in gpurun_out/ln/*.log; do echo "== $f"; cat $f; done
