set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|[A-Za-z]*Mfma[A-Za-z]*)\b" | sort -u > $R/gpurun_out/counters.txt || true
wc -l $R/gpurun_out/counters.txt
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o p -- python3 $R/scripts/prof_conv.py 64 112 128 128 3 1 3 fwd > $R/gpurun_out/pmc1.log 2>&1 || tail -5 $R/gpurun_out/pmc1.log
ls $R/gpurun_out/pmc1
