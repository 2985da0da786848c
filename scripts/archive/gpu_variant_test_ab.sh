# A library variant (scripts/build_variant_lib.sh <name> -D...): conv kernel tests FIRST (a wrong kernel must not be timed), then the
# same-box A/B in the full step:  bash scripts/gpu_variant_test_ab.sh <name>
set -e
V=$1
mkdir -p gpurun_out/$V
SGG_HIP_LIB=scene-graph-gan_amd/_prof/libsgg_hip_$V.so timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_conv_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/$V/test.log 2>&1 || { tail -30 gpurun_out/$V/test.log; exit 1; }
tail -2 gpurun_out/$V/test.log
bash scripts/gpu_ab.sh $V base $V
