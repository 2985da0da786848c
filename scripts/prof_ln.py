"""LayerNorm + ELU streaming kernels alone at the configs[1] tensor sizes: microseconds and algorithmic TB/s per call
(HIP events around 20 calls).  SGG_HIP_LIB selects the library build (scripts/build_variant_lib.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa: E402,F401
from sgg_amd.lib import HipKernels  # noqa: E402


_FLUSH = None


def timeit(fn, n=10):
    """Mean microseconds of fn(); 1 GB of unrelated traffic before every call, so nothing of the previous call is still cached."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(256 << 20, device="cuda")
    fn()
    tot = 0.0
    for _ in range(n):
        _FLUSH.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


def main():
    K = HipKernels("cuda:0")
    tot = 0.0
    for shape in [(64, 224, 224, 32), (64, 112, 112, 64), (64, 112, 112, 128), (64, 56, 56, 256), (64, 28, 28, 512)]:
        B, H, W, C = shape
        y = torch.randn(shape, device="cuda")
        da = torch.randn(shape, device="cuda")
        a, dy = torch.empty_like(y), torch.empty_like(y)
        gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        st = torch.empty((B, 2), device="cuda")
        dg, db, dbias = (torch.empty(C, device="cuda") for _ in range(3))
        amax = torch.zeros(2, device="cuda")
        nb = y.numel() * 4.0
        # tile statistics as the conv epilogue would deliver them: take the kernel's own partials layout via a first full call
        t_full = timeit(lambda: K.ln_elu_fwd(y, gamma, beta, a, st, amax[0:1]))
        ts = torch.zeros((B, 8, 4), device="cuda")
        ts[:, :, 0] = H * W * C / 8.0
        ts[:, :, 2] = H * W * C / 8.0
        t_apply = timeit(lambda: K.ln_elu_fwd(y, gamma, beta, a, st, amax[0:1], ts))
        t_bwd = timeit(lambda: K.ln_elu_bwd(y, da, gamma, beta, st, dy, dg, db, dbias, amax[1:2]))
        print("%-20s fwd(stats+apply) %7.1f us %5.2f TB/s | apply %7.1f us %5.2f TB/s | bwd %7.1f us %5.2f TB/s" % (
            "x".join(map(str, shape)), t_full, 3 * nb / t_full / 1e6, t_apply, 2 * nb / t_apply / 1e6, t_bwd, 5 * nb / t_bwd / 1e6))
        tot += t_apply + t_bwd
    print("sum(apply + bwd) over the five shapes: %.1f us" % tot)


if __name__ == "__main__":
    main()
