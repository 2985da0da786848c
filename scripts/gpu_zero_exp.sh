cd $GRAFT_REPO_ROOT
for z in 0 1 2; do
echo "SGG_PROF_ZERO=$z"
SGG_PROF_ZERO=$z timeout -k 10 120 python scripts/prof_conv.py 64 112 128 128 3 1 20 fwd_ws
SGG_PROF_ZERO=$z SGG_CONV_HALO=0 timeout -k 10 120 python scripts/prof_conv.py 64 112 128 128 3 1 20 fwd_ws
done
