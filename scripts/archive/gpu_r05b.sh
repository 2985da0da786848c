# round 5: persistent convolution kernels on 30 / 28 of an XCD's 32 CUs (the rest left to the other streams' HBM-bound kernels): full-step A/B
set -e
{
echo "# two-stream schedule, batch 64 / 224x224 / vocab 1000, 10 timed steps, interleaved; base = 32 CUs per XCD (in tree)"
bash scripts/gpu_ab.sh cuab base cu30 cu28
} > gpurun_out/r05_persistent_cu_cap_ab.log 2>&1
cat gpurun_out/r05_persistent_cu_cap_ab.log
