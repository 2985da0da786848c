"""LayerNorm + ELU backward of one layer (reduction pass + apply pass = 4 tensor reads + 1 write) on the whole batch against the same
call on CHUNKS of the batch: does the apply pass of a chunk find its y / dA in the 256 MiB Infinity Cache when it runs right behind
that chunk's reduction pass?   python scripts/ubench/ln_bwd_chunked.py
(timing probe: the chunks get amax words of their own, which the product could not do - see DESIGN.md section 9)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd.lib import HipKernels  # noqa: E402

dev = torch.device("cuda:0")
K = HipKernels(dev)
g = torch.Generator(device="cpu").manual_seed(0)


def bench(B, H, W, C, chunks, reps=10, s16=True):
    y = torch.randn(B, H, W, C, generator=g).to(dev)
    da = torch.randn(B, H, W, C, generator=g).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    stats = torch.stack([y.mean(dim=(1, 2, 3)), 1.0 / y.std(dim=(1, 2, 3))], dim=1).contiguous()
    dy = torch.empty_like(y)
    need = K.lib.sgg_layernorm_hwc_elu_workspace_bytes(B, H * W, C)
    out = {}
    for nch in chunks:
        n = B // nch
        ws = [torch.zeros(need, dtype=torch.uint8, device=dev) for _ in range(nch)]
        am = torch.zeros(nch, device=dev)
        pq = torch.zeros(nch, 2, device=dev)

        def run():
            for c in range(nch):
                s = slice(c * n, (c + 1) * n)
                K.ln_elu_bwd(y[s], da[s], gamma, beta, stats[s], dy[s], None, None, None, am[c:c + 1] if s16 else None, None, ws[c], s16,
                             pq[c] if s16 else None)
        for _ in range(2):
            pq.zero_()
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pq.zero_()
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        out[nch] = us
        print("  B %d %dx%dx%d (%.0f MB per tensor), %2d chunk(s) of %2d samples: %7.1f us  = %.2f TB/s of 5 tensor passes" %
              (B, H, W, C, 4e-6 * y.numel(), nch, n, us, 5 * 4 * y.numel() / us / 1e6), flush=True)
    return out


print("pre-split dy (the product's default):")
bench(64, 112, 112, 128, (1, 2, 4, 8, 16))
bench(64, 112, 112, 64, (1, 2, 4, 8))
bench(64, 56, 56, 256, (1, 2, 4))
bench(64, 224, 224, 32, (1, 4, 8, 16))
