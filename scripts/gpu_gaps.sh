set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-rows 0 --no-kernel-timing > $O/bench.json 2> $O/err.log
python3 - <<'PY'
import csv, os
f = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/gaps/k_kernel_trace.csv")
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# last 3 steps ~ last 60% of the trace: take the final 3/5 of kernels by time window
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - 0.45 * (t1 - t0)
sel = [r for r in rows if r[0] >= cut]
busy = 0; gaps = []; last_end = sel[0][0]
small = 0
for s, e, n in sel:
    if s > last_end: gaps.append(s - last_end)
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
    if e - s < 20000: small += 1
span = sel[-1][1] - sel[0][0]
g2 = [g for g in gaps if g < 100000]
print("gaps below 100 us: n=%d total %.2f ms mean %.2f us; per kernel %.2f us over %d kernels" % (len(g2), sum(g2) / 1e6, sum(g2) / max(len(g2), 1) / 1e3, sum(g2) / len(sel) / 1e3, len(sel)))
print("window %.2f ms, kernels %d (shorter than 20 us: %d), busy %.2f ms, idle %.2f ms (%.1f %%), gaps: n=%d mean %.2f us" % (
    span / 1e6, len(sel), small, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, len(gaps), (sum(gaps) / max(len(gaps), 1)) / 1e3))
PY
rm -f $O/k_kernel_trace.csv
