# round 3, first GPU call: the new f1 / f3 / concurrency tests first, then the whole GPU suite, then the N > 1 self-launch path of
# bench.py (2 ranks share the GPU over gloo) and the parity-enforcing exit codes.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_f1_f3_gpu.py tests/test_api_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x -s > $O/new_tests.log 2>&1 || { tail -60 $O/new_tests.log; exit 1; }
tail -3 $O/new_tests.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
SGG_DP_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --size 64 --vocab 50 > $O/bench_gpus2.json 2> $O/bench_gpus2.err || { tail -30 $O/bench_gpus2.err; exit 1; }
head -c 600 $O/bench_gpus2.json; echo
set +e
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --batch 8 --size 64 --vocab 50 --conv-precision 1 --f32-steps 1 --ci10-steps 0 --two-stream-steps 0 > $O/bench_prec1.json 2> $O/bench_prec1.err
echo "precision 1 exit code: $?" | tee $O/bench_prec1.rc
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --batch 8 --size 64 --vocab 50 --f32-steps 1 --ci10-steps 0 --two-stream-steps 0 > $O/bench_small.json 2> $O/bench_small.err
echo "default precision exit code: $?" | tee $O/bench_small.rc
