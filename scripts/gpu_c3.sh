set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "c3 or conv_fwd_dgrad_wgrad or epilogue" 2>&1 | tail -3
timeout -k 10 120 python scripts/prof_conv.py 64 224 3 32 3 1 10 fwd
timeout -k 10 120 python scripts/prof_conv.py 64 224 3 32 3 1 10 wgrad
