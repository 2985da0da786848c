"""Timing of the hoisted attention product (sgg_attn_ctx_gemm_{fwd,dgrad,wgrad}) at configs[1]: B rows x K = 100352 x L = 196."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgg_amd  # noqa: F401
from sgg_amd.lib import HipKernels

K = HipKernels("cuda:0")
L, C = 196, 512
for B in (64, 192):
    ctx = torch.randn((B, L * C), device="cuda")
    W = torch.randn((L * C, L), device="cuda") * 0.01
    bias = torch.randn(L, device="cuda")
    P = torch.empty((B, L), device="cuda")
    dP = torch.randn((B, L), device="cuda")
    dctx = torch.zeros((B, L * C), device="cuda")
    dW = torch.zeros((L * C, L), device="cuda")
    for name, f, nbytes in (("fwd", lambda: K.attn_ctx_fwd(ctx, W, bias, P), 4.0 * (ctx.numel() + W.numel())),
                            ("dgrad", lambda: K.attn_ctx_dgrad(dP, W, dctx, accumulate=True), 4.0 * (2 * dctx.numel() + W.numel())),
                            ("wgrad", lambda: K.attn_ctx_wgrad(ctx, dP, dW, accumulate=True), 4.0 * (ctx.numel() + 2 * W.numel()))):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("attn_ctx %-5s B=%3d: %6.1f us  %5.2f TB/s algorithmic" % (name, B, us, nbytes / us / 1e6))
