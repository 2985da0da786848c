# per-layer wgrad rates at configs[1] (kernel + slab reduce per call): gpurun_out/<tag>/wgrad.log
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-wg}
mkdir -p $O
for spec in "64 224 32 32 3 1" "64 224 32 32 5 2" "64 112 32 64 3 1" "64 112 64 64 3 1" "64 112 64 128 3 1" "64 112 128 128 3 1" "64 112 128 128 5 2" "64 56 128 256 3 1" "64 56 256 256 3 1" "64 56 256 512 5 2" "64 28 512 512 5 2"; do
  set -- $spec
  echo -n "wgrad  " | tee -a $O/wgrad.log
  timeout -k 10 120 python scripts/prof_conv.py $1 $2 $3 $4 $5 $6 10 wgrad 2>&1 | tail -1 | tee -a $O/wgrad.log
done
