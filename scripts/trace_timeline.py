"""Where a two-stream step spends its time: python scripts/trace_timeline.py <kernel_trace.csv> [n_last_steps] [--gated] [--gantt]
(rocprofv3 --kernel-trace --output-format csv of `bench.py --steps N ...`; --gated: of scripts/trace_step.py, whose steps are
delimited by GPU-side spin kernels - the window is the LAST step, from the end of its spin to its last kernel; --gantt: also list
every kernel of the last step longer than 15 us with its queue).  Kernels are classed as MFMA-bound convolution work or
"other" (LayerNorm, heads, optimiser, ...); the union of their [start, end) intervals over the last steps of the trace gives the time
with a matrix kernel resident, with only other kernels resident, and with nothing resident, and the kernels that fill the
matrix-idle time."""
import csv
import sys
from collections import defaultdict

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
gated, gantt = "--gated" in sys.argv, "--gantt" in sys.argv
path = argv[0]
rows = []
import gzip
for r in csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
mfma = lambda n: n.startswith("void conv_") and "c3" not in n and "slab_reduce" not in n
# the timed window: from the first Adam launch of the last `nsteps` steps ... simply the last fraction of the trace by Adam launches
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
nsteps = int(argv[1]) if len(argv) > 1 else 3
lo = rows[adam[-2 * nsteps - 1]][1] if len(adam) > 2 * nsteps else rows[0][0]
hi = rows[adam[-1]][1]
if gated:
    spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[2]]
    nsteps, lo = 1, rows[spins[-1]][1]
    hi = max(r[1] for r in rows[spins[-1] + 1:])
win = [r for r in rows if r[0] >= lo and r[1] <= hi and "spin_kernel" not in r[2]]


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(u):
    return sum(b - a for a, b in u)


um = union([(a, b) for a, b, n, q in win if mfma(n)])
ua = union([(a, b) for a, b, n, q in win])
total = hi - lo
print("window %.2f ms = %d steps of %.2f ms; kernels %d, queues %s" % (total / 1e6, nsteps, total / 1e6 / nsteps, len(win), sorted({q for *_, q in win})))
print("  a matrix (conv) kernel resident : %6.2f ms per step" % (length(um) / 1e6 / nsteps))
print("  only other kernels resident     : %6.2f ms per step" % ((length(ua) - length(um)) / 1e6 / nsteps))
print("  nothing resident                : %6.2f ms per step" % ((total - length(ua)) / 1e6 / nsteps))
# who runs while no matrix kernel is resident
acc = defaultdict(float)
j = 0
for a, b, n, q in win:
    if mfma(n):
        continue
    # part of [a, b) outside um
    out = b - a
    for c, d in um:
        if d <= a:
            continue
        if c >= b:
            break
        out -= min(b, d) - max(a, c)
    acc[n.split("(")[0][:70]] += max(out, 0)
print("  kernels resident while no matrix kernel is (ms per step, summed per kernel; overlapping each other is counted twice):")
for n, v in sorted(acc.items(), key=lambda kv: -kv[1])[:25]:
    print("    %-72s %6.3f" % (n, v / 1e6 / nsteps))

# ---- the matrix-idle intervals of ONE step (the last one), longest first: where in the schedule the matrix pipe has nothing -------
step_lo = lo if gated else (rows[adam[-3]][1] if len(adam) > 3 else lo)
gaps = []
prev = step_lo
for c, d in um:
    if d <= step_lo:
        continue
    if c > prev:
        gaps.append((prev, c))
    prev = max(prev, d)
if hi > prev:
    gaps.append((prev, hi))
print("  matrix-idle intervals of the last step (%.2f ms in %d intervals); the 14 longest, in time order, with the kernels inside:"
      % (sum(b - a for a, b in gaps) / 1e6, len(gaps)))
top = sorted(sorted(gaps, key=lambda g: g[0] - g[1])[:14])
for a, b in top:
    inside = defaultdict(lambda: [0, 0.0])
    for s, e, n, q in win:
        if e > a and s < b and not mfma(n):
            k = n.replace("void ", "").split("(")[0][:44]
            inside[k][0] += 1
            inside[k][1] += (min(e, b) - max(s, a)) / 1e3
    busy = length(union([(max(s, a), min(e, b)) for s, e, n, q in win if e > a and s < b and not mfma(n)])) / 1e3
    # the matrix kernels on either side
    before = [n for s, e, n, q in win if mfma(n) and e <= a + 1]
    after = [n for s, e, n, q in win if mfma(n) and s >= b - 1]
    name = lambda n: n.replace("void ", "").split("(")[0][:40]
    print("    +%7.3f ms  %7.1f us (busy %6.1f)  after %-40s before %-40s : %s"
          % ((a - step_lo) / 1e6, (b - a) / 1e3, busy, name(before[-1]) if before else "-", name(after[0]) if after else "-",
             ", ".join("%s x%d %.0fus" % (k, v[0], v[1]) for k, v in sorted(inside.items(), key=lambda kv: -kv[1][1])[:5])))

if gantt:
    print("  kernels of the last step longer than 15 us (start, end in ms from the step's start; us; queue):")
    for a, b, n, q in win:
        if a >= step_lo and b - a > 15000:
            print("    %8.3f %8.3f %6.0f q%s %s" % ((a - step_lo) / 1e6, (b - step_lo) / 1e6, (b - a) / 1e3, q, n.replace("void ", "").split("(")[0][:64]))
