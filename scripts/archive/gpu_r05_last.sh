# round 5: last check of the committed tree: smoke + the GPU suite
set -e
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.log 2>&1 || { tail -20 gpurun_out/r05/smoke.log; exit 1; }
tail -1 gpurun_out/r05/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r05/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r05/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r05/pytest_gpu.log
