# round 5: per-shape kernel rates with / without the four-block producer / consumer tiles (serial steps, HIP events per launch)
set -e
for v in "" "halo_pc64=0"; do
  SGG_OPTIONS="$v" timeout -k 10 300 python bench.py --steps 3 --warmup 2 --cpu-rows 0 --f32-steps 0 --ci10-steps 0 --other-configs 0 --bitwise-iters 0 --serial-steps 3 --per-shape > gpurun_out/pc64_shape_$([ -z "$v" ] && echo on || echo off).json 2> /dev/null
done
python - <<'PY'
import json
for tag in ("on", "off"):
    p = json.loads(open("gpurun_out/pc64_shape_%s.json" % tag).read().strip().splitlines()[-1])
    print("#", tag, "serial %.2f ms" % p["serial"]["ms_per_step"])
    for k, v in p["per_shape"].items():
        if ("halo3_pc_kernel" in k and k.split("<")[1].split(">")[0].endswith("4")) or "conv_halo3_kernel<2,64" in k or "conv_halo3_pc" in k:
            print("   %-70s %s" % (k, v))
PY
