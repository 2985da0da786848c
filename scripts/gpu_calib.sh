# Counter calibration on the GPU box: bash scripts/gpu_calib.sh   -> gpurun_out/r05_counter_calibration.log
# Two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domain beside --pmc).
set -e
R=$GRAFT_REPO_ROOT
BIN=$R/scripts/ubench/counter_calib.bin
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o $BIN $R/scripts/ubench/counter_calib.hip
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/calib_f $R/gpurun_out/calib_w
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib_f -o f -- $BIN > $R/gpurun_out/calib_f.log 2>&1 || tail -5 $R/gpurun_out/calib_f.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib_w -o w -- $BIN > $R/gpurun_out/calib_w.log 2>&1 || tail -5 $R/gpurun_out/calib_w.log
python3 - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r05_counter_calibration.log
import csv, glob, os
from collections import defaultdict
R = os.environ["GRAFT_REPO_ROOT"]
BYTES = 64 * 112 * 112 * 128 * 4
print("# scripts/ubench/counter_calib.hip under rocprofv3 --pmc (two passes); every kernel moves %d bytes exactly once; 3 launches each" % BYTES)
print("# factor = counter KiB * 1024 / bytes moved (1.00 = the counter reads the bytes; 0.50 = it reads half of them)")
for d, cname in (("calib_f", "FETCH_SIZE"), ("calib_w", "WRITE_SIZE")):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(R, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-11s %-34s launches %d  KiB per launch %s  factor %s" % (cname, k, len(v), ["%.0f" % x for x in v], ["%.3f" % (x * 1024 / BYTES) for x in v]))
print("# kernel durations (kernel trace of the FETCH_SIZE pass): the same bytes in how many microseconds")
dur = defaultdict(list)
for f in glob.glob(os.path.join(R, "gpurun_out", "calib_f", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items()):
    if k.startswith(("r_", "w_", "void w_")):
        med = sorted(v)[len(v) // 2]
        print("%-34s us %s   %.2f TB/s" % (k, ["%.1f" % x for x in v], BYTES / med / 1e6))
PY
