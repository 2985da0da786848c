"""Build libsgg_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The shared object is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB_PATH = os.path.join(HERE, "libsgg_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed-fp32 VALU instructions (v_pk_fma_f32 ...): a wave issuing them beside another stream's MFMA kernel on the same SIMD
# was observed to return wrong low halves in lanes 48-63 (conv_c3_fwd / conv_c3_wgrad beside conv_s2_kernel: 116 of 120 outputs
# differed; tests/test_concurrency_gpu.py); without them 0 of 120, and the step is 0.4 ms faster.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"] + NO_PACKED_FP32


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
    cmd = [HIPCC] + FLAGS + ["-I", CSRC, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    # (the host pass of hipcc does not know the device feature and says so: not a diagnostic of ours)
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)
    return obj, err


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    stamp = os.path.join(OUT_DIR, "stamp")
    dig = _digest(srcs + headers)
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB_PATH
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    if verbose:
        for _, err in results:
            if err.strip():
                print(err, file=sys.stderr)
    objs = [o for o, _ in results]
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
