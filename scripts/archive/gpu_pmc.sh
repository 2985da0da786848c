# One conv shape under rocprofv3 PMC counters (two passes): bash scripts/gpu_pmc.sh <precision> <mode> "<B H Cin Cout k s>"
#   e.g. bash scripts/gpu_pmc.sh 2 fwd_ws "64 56 256 512 5 2"      (modes: scripts/prof_conv.py)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PREC=${1:-2}
MODE=${2:-fwd_ws}
SHAPE=${3:-64 112 128 128 3 1}
SGG_CONV_PRECISION=$PREC timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 $MODE > $R/gpurun_out/pmc1.log 2>&1 || tail -5 $R/gpurun_out/pmc1.log
SGG_CONV_PRECISION=$PREC timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 $MODE > $R/gpurun_out/pmc2.log 2>&1 || tail -5 $R/gpurun_out/pmc2.log
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc1", "pmc2"):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for f in glob.glob(os.path.join(R, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "conv" in k:
            print(d, k, {c: "%.3g" % x for c, x in v.items()})
PY
