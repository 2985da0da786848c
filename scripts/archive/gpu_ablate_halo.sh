# Where does conv_halo3_kernel<2,128,...> lose its time?  (1) PMC counters of the plain kernel on the conv2_4 / conv3_2 shapes;
# (2) timing-only ablation builds (-DSGG_ABL_*: wrong results, same MFMA stream) on the same box, interleaved repetitions.
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-abl}
mkdir -p $O
# (the variant libraries are built in the container beforehand: scripts/build_variant_one.sh abl_<v> conv_halo.hip -DSGG_ABL_...)
for rep in 1 2; do
  for v in base nob noa nostage noepi noall; do
    if [ "$v" = base ]; then unset SGG_HIP_LIB; else export SGG_HIP_LIB=$R/scene-graph-gan_amd/_prof/libsgg_hip_abl_$v.so; fi
    for shape in "64 112 128 128 3 1" "64 56 256 256 3 1"; do
      echo -n "$v rep $rep: " | tee -a $O/times.log
      timeout -k 10 120 python scripts/prof_conv.py $shape 20 fwd_ws 2>&1 | tail -1 | tee -a $O/times.log
    done
  done
done
unset SGG_HIP_LIB
cd /tmp && export TMPDIR=/tmp
SHAPE="64 112 128 128 3 1"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc1 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 fwd_ws > $O/pmc1.log 2>&1 || tail -5 $O/pmc1.log
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc2 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 fwd_ws > $O/pmc2.log 2>&1 || tail -5 $O/pmc2.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc3 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 fwd_ws > $O/pmc3.log 2>&1 || tail -5 $O/pmc3.log
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc4 -o p -- python3 $R/scripts/prof_conv.py $SHAPE 3 fwd_ws > $O/pmc4.log 2>&1 || tail -5 $O/pmc4.log
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
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
