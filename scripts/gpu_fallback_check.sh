cd $GRAFT_REPO_ROOT
SGG_CONV_HALO=0 timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "not halo" 2>&1 | tail -2
SGG_CONV_PRECISION=3 timeout -k 10 500 python -m pytest tests/test_step_gpu.py tests/test_api_gpu.py -m gpu -q -x 2>&1 | tail -2
SGG_CONV_PRECISION=0 timeout -k 10 500 python -m pytest tests/test_step_gpu.py -m gpu -q -x 2>&1 | tail -2
