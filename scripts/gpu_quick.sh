set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
bash scripts/gpu_bench_short.sh
