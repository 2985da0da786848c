# timing-only ablations of the producer / consumer kernel (+ PMC counters of the real one)
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-pc3}
mkdir -p $O
for rep in 1 2; do
  for v in $VARIANTS; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_$v.so; fi
    for spec in "64 112 128 128 3 1 fwd_ws" "64 56 256 256 3 1 fwd_ws"; do
      set -- $spec
      echo -n "$v rep $rep: " | tee -a $O/times.log
      timeout -k 10 120 python scripts/prof_conv.py $1 $2 $3 $4 $5 $6 20 $7 2>&1 | tail -1 | tee -a $O/times.log
    done
  done
done
unset SGG_HIP_LIB
cd /tmp && export TMPDIR=/tmp
SHAPE="64 112 128 128 3 1"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc1 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 fwd_ws > $O/pmc1.log 2>&1 || tail -5 $O/pmc1.log
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $O/pmc2 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 fwd_ws > $O/pmc2.log 2>&1 || tail -5 $O/pmc2.log
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
for d in ("pmc1", "pmc2"):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "conv_halo" in k:
            print(d, k, {c: "%.4g (n=%d)" % (x, n[(k, c)]) for c, x in v.items()})
    for f in glob.glob(os.path.join(O, d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_halo" in r["Kernel_Name"]:
                print("  ", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"))
PY
