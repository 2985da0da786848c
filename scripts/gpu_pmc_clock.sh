set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for halo in 1 0; do
SGG_CONV_HALO=$halo timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/clk$halo -o p -- python3 $R/scripts/prof_conv.py 64 112 128 128 3 1 40 fwd_ws > $R/gpurun_out/clk$halo.log 2>&1 || tail -5 $R/gpurun_out/clk$halo.log
done
