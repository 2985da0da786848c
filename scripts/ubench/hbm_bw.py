"""Achievable HBM bandwidth on this box (torch copy / read-only reduce) next to the LayerNorm kernels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sgg_amd
from sgg_amd.lib import HipKernels
K = HipKernels("cuda:0")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (B, H, C) in ((64, 224, 32), (64, 112, 128), (64, 56, 256)):
    x = torch.randn((B, H, H, C), device="cuda"); y = torch.empty_like(x); d = torch.randn_like(x); dy = torch.empty_like(x)
    nb = x.numel() * 4
    ms = t(lambda: y.copy_(x)); print("B%d H%d C%d  %4.0f MB  copy        %.1f us  %.2f TB/s" % (B, H, C, nb / 1e6, ms * 1e3, 2 * nb / ms / 1e9))
    ms = t(lambda: x.sum()); print("                       reduce      %.1f us  %.2f TB/s" % (ms * 1e3, nb / ms / 1e9))
    g, b_ = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"); st = torch.empty((B, 2), device="cuda")
    ms = t(lambda: K.ln_elu_fwd(x, g, b_, y, st)); print("                       ln fwd (stats+apply, 3 passes) %.1f us  %.2f TB/s" % (ms * 1e3, 3 * nb / ms / 1e9))
    gg, gb, gbias = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ms = t(lambda: K.ln_elu_bwd(x, d, g, b_, st, dy, gg, gb, gbias)); print("                       ln bwd (5 passes) %.1f us  %.2f TB/s" % (ms * 1e3, 5 * nb / ms / 1e9))
