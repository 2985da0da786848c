"""Copy the judged summaries of one gpurun profile directory (scripts/gpu_round.sh <step> <tag>) into profiles/ and derive the
roofline `traffic` values.   python scripts/summarize_profiles.py gpurun_out/r02 r02"""
import csv, json, os, shutil, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
if os.path.exists(os.path.join(src, "stats", "k_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "stats", "k_kernel_stats.csv"), os.path.join(dst, "%s_rocprofv3_kernel_stats.csv" % tag))
for f in ("bench.json", "bench_under_rocprof.json", "bench_config3.json", "bench_config4.json", "pytest_gpu.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


# Calibration (scripts/ubench/counter_calib.hip, profiles/r05_counter_calibration.log): FETCH_SIZE reads 0.500 of the bytes of
# 16-byte-per-lane reads - global_load, contiguous `buffer_load ... lds` AND the per-lane-offset (gather) `buffer_load ... lds` of the
# patch DMAs alike (0.502) -> doubled.  WRITE_SIZE reads 1.000 of default-policy stores and 1.005 of nontemporal stores that cover
# whole 128-byte lines, but 1.27 of NONTEMPORAL 16-byte stores that cover a 64-byte half line per instruction (the producer /
# consumer kernel's epilogue: 16-column accumulator tiles): for those kernels the calibrated write bytes are WRITE_SIZE / 1.27.
WRITE_FACTOR = {"conv_halo3_pc_kernel": 1.27}

fpath, wpath = os.path.join(src, "pmc_fetch", "f_counter_collection.csv"), os.path.join(src, "pmc_write", "w_counter_collection.csv")
if os.path.exists(fpath) and os.path.exists(wpath):
    fetch, write = per_kernel(fpath, "FETCH_SIZE"), per_kernel(wpath, "WRITE_SIZE")
    rows, traffic = [], {}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n, f = fetch[k]
        w = write.get(k, [0, 0.0])[1]
        short = k.replace("void ", "").split("(")[0].replace(" ", "")
        wf = WRITE_FACTOR.get(short.split("<")[0], 1.0)
        read_b, write_raw = 2.0 * f * 1024.0 / max(n, 1), w * 1024.0 / max(n, 1)
        per_launch = read_b + write_raw / wf
        rows.append({"kernel": k, "launches": n, "fetch_kib_raw_sum": f, "write_kib_sum": w, "read_bytes_per_launch": read_b,
                     "write_bytes_per_launch_raw": write_raw, "write_counter_factor": wf, "write_bytes_per_launch": write_raw / wf,
                     "hbm_bytes_per_launch_corrected": per_launch})
        traffic[short] = per_launch
    json.dump(rows, open(os.path.join(dst, "%s_pmc_hbm_traffic.json" % tag), "w"), indent=1)
    json.dump(traffic, open(os.path.join(dst, "roofline_traffic.json"), "w"), indent=1)
print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(tag) or f == "roofline_traffic.json"))
