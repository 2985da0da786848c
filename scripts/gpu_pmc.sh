set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PREC=${1:-6}
MODE=${2:-fwd}
SGG_CONV_PRECISION=$PREC timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o p -- python3 $R/scripts/prof_conv.py 64 112 128 128 3 1 3 $MODE > $R/gpurun_out/pmc1.log 2>&1 || tail -5 $R/gpurun_out/pmc1.log
SGG_CONV_PRECISION=$PREC timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o p -- python3 $R/scripts/prof_conv.py 64 112 128 128 3 1 3 $MODE > $R/gpurun_out/pmc2.log 2>&1 || tail -5 $R/gpurun_out/pmc2.log
