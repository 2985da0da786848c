cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_full.log 2>&1
tail -5 gpurun_out/pytest_gpu_full.log
