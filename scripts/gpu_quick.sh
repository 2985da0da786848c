set -e
cd $GRAFT_REPO_ROOT
for prec in 6 3; do
for args in "64 112 128 128 3 1 5 fwd" "64 56 256 512 5 2 5 fwd" "64 28 512 512 5 2 5 fwd"; do
  SGG_CONV_PRECISION=$prec timeout -k 10 120 python scripts/prof_conv.py $args 2>/dev/null
done
done
