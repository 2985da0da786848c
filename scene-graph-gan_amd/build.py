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
    # (the x86 host pass of hipcc does not know the device feature and says so; whether the DEVICE pass still honours it is checked on
    # the linked code objects themselves: check_no_packed_fp32)
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)
    return obj, err


OBJDUMP = os.environ.get("LLVM_OBJDUMP", os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib", "llvm", "bin",
                                                      "llvm-objdump"))


M0_ASM_KERNELS = ("conv_wgrad_dma", "conv_s2_kernel")      # kernels whose inline assembly writes M0 (csrc/conv_wgrad_dma.hip, conv_s2.hip)


def check_device_code(lib_path: str = LIB_PATH) -> int:
    """Disassemble every gfx950 code object of the linked library and verify two properties the sources can only ask for:

    1. no packed-fp32 VALU instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, ...) is present.  The correctness of running these
       kernels beside another stream's MFMA kernels (train.py's two-stream default, RCCL kernels in data-parallel runs, the loader's
       resize stream) rests on their absence (NO_PACKED_FP32 above); the compiler flag that removes them is an internal target
       feature, so the build verifies its EFFECT instead of trusting the flag;
    2. in the kernels whose inline assembly sets M0 for `buffer_load ... lds` (M0_ASM_KERNELS) nothing else uses M0: every
       instruction naming m0 is that assembly's own `s_mov_b32 m0, sN`, and no movrel / readlane-by-M0 / sendmsg / GWS instruction
       (the implicit M0 readers) occurs.  M0 is a reserved register of the target (it cannot go into the asm's clobber list:
       -Winline-asm), so a compiler that started to keep a value in it inside these kernels would corrupt LDS destinations
       silently - this check turns that into a build failure.

    Returns the number of device instructions inspected."""
    import re
    import shutil
    tmp = os.path.join(OUT_DIR, "disasm")
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    copy = os.path.join(tmp, "lib.so")
    shutil.copy(lib_path, copy)
    if not os.path.exists(OBJDUMP):
        raise RuntimeError("llvm-objdump not found at %s (set LLVM_OBJDUMP): the build cannot verify that the library is free of "
                           "packed-fp32 VALU instructions" % OBJDUMP)
    r = subprocess.run([OBJDUMP, "--offloading", copy], capture_output=True, text=True)     # writes lib.so.<i>.<triple> next to `copy`
    if r.returncode != 0:
        raise RuntimeError("llvm-objdump --offloading failed:\n%s" % r.stderr)
    objs = sorted(f for f in os.listdir(tmp) if f.endswith(ARCH))
    if not objs:
        raise RuntimeError("no %s code object found in %s" % (ARCH, lib_path))
    pat, n_ins, bad = re.compile(r"\bv_pk_[a-z0-9]+_f32\b"), 0, []
    sym_pat = re.compile(r"^[0-9a-f]+ <(\S+)>:")
    m0_ok = re.compile(r"^\s*s_mov_b32 m0, s\d+\b")
    m0_implicit = re.compile(r"\b(v_movrel[sd]*_b32|s_movrel[sd]_b(32|64)|v_readlane_b32 \S+, \S+, m0|v_writelane_b32 \S+, \S+, m0|s_sendmsg|ds_gws_\w+|"
                             r"v_interp_\w+|s_load_\w+ .*\bm0\b)")
    m0_bad, m0_seen, cur = [], 0, ""
    for f in objs:
        d = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True)
        if d.returncode != 0:
            raise RuntimeError("llvm-objdump -d failed on %s:\n%s" % (f, d.stderr))
        for line in d.stdout.splitlines():
            m = sym_pat.match(line)
            if m:
                cur = m.group(1)
                continue
            if "\t" in line:
                n_ins += 1
                if pat.search(line):
                    bad.append("%s: %s" % (f, line.strip()))
                if any(k in cur for k in M0_ASM_KERNELS):
                    ins = line.split("//")[0]
                    if re.search(r"\bm0\b", ins):
                        m0_seen += 1
                        if not m0_ok.match(ins):
                            m0_bad.append("%s: %s" % (cur, ins.strip()))
                    elif m0_implicit.search(ins):
                        m0_bad.append("%s: %s" % (cur, ins.strip()))
    shutil.rmtree(tmp, ignore_errors=True)
    if bad:
        raise RuntimeError("libsgg_hip.so contains %d packed-fp32 VALU instruction(s); the build flag %s no longer removes them:\n%s"
                           % (len(bad), " ".join(NO_PACKED_FP32), "\n".join(bad[:10])))
    if m0_bad:
        raise RuntimeError("a kernel that sets M0 in inline assembly has another M0 user (%d instruction(s)):\n%s"
                           % (len(m0_bad), "\n".join(m0_bad[:10])))
    if m0_seen == 0:
        raise RuntimeError("no `s_mov_b32 m0` found in %s: the M0 check looked at the wrong symbols" % (M0_ASM_KERNELS,))
    if n_ins < 1000:
        raise RuntimeError("disassembly of %s looks empty (%d instructions)" % (lib_path, n_ins))
    return n_ins


check_no_packed_fp32 = check_device_code      # (the name earlier rounds and DESIGN.md use)


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
    check_device_code(LIB_PATH)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
