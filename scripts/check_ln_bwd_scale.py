"""One-pass LayerNorm backward at growing sizes against an fp64 evaluation of the same formula on the device:
python scripts/check_ln_bwd_scale.py B H W C.  Prints the worst error, the bounded-wait flag and microseconds per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa: E402,F401
from sgg_amd.lib import HipKernels  # noqa: E402


def main():
    B, H, W, C = map(int, sys.argv[1:5])
    K = HipKernels("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(3)
    shape = (B, H, W, C)
    y = torch.randn(shape, device="cuda", generator=g) * 2.0 + 0.3
    da = torch.randn(shape, device="cuda", generator=g)
    gamma = 1.0 + 0.2 * torch.randn(C, device="cuda", generator=g)
    beta = 0.2 * torch.randn(C, device="cuda", generator=g)
    a, dy = torch.empty_like(y), torch.empty_like(y)
    st = torch.empty((B, 2), device="cuda")
    dg, db, dbias = (torch.empty(C, device="cuda") for _ in range(3))
    K.ln_elu_fwd(y, gamma, beta, a, st)
    K.ln_elu_bwd(y, da, gamma, beta, st, dy, dg, db, dbias)
    torch.cuda.synchronize()
    flag = K.ln_bwd_timed_out(shape)
    worst = 0.0
    dg_ref = torch.zeros(C, device="cuda", dtype=torch.float64)
    db_ref = torch.zeros_like(dg_ref)
    for b in range(B):
        mean, rstd = st[b, 0].double(), st[b, 1].double()
        xh = (y[b].double() - mean) * rstd
        n = xh * gamma.double() + beta.double()
        dn = da[b].double() * torch.where(n > 0, torch.ones_like(n), torch.exp(n))
        dxh = dn * gamma.double()
        ref = rstd * (dxh - dxh.mean() - xh * (dxh * xh).mean())
        worst = max(worst, float((dy[b].double() - ref).abs().max() / ref.abs().max()))
        dg_ref += (dn * xh).sum(dim=(0, 1))
        db_ref += dn.sum(dim=(0, 1))
    eg = float((dg.double() - dg_ref).abs().max() / dg_ref.abs().max())
    eb = float((db.double() - db_ref).abs().max() / db_ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.ln_elu_bwd(y, da, gamma, beta, st, dy, dg, db, dbias)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    ws = torch.empty(K.ln_workspace_bytes(shape), dtype=torch.uint8, device="cuda")
    e0.record()
    for _ in range(10):
        K.ln_elu_bwd(y, da, gamma, beta, st, dy, None, None, None, ws=ws)      # (deferred parameter gradients: the streaming part alone)
    e1.record()
    torch.cuda.synchronize()
    us_d = e0.elapsed_time(e1) * 100.0
    nb = y.numel() * 4.0
    print("%s: dy err %.2e dgamma %.2e dbeta %.2e timed_out %s  %.1f us/call, deferred %.1f us  %.2f TB/s (3 passes) %s" % (
        "x".join(map(str, shape)), worst, eg, eb, flag, us, us_d, 3 * nb / us_d / 1e6, os.environ.get("SGG_HIP_LIB", "")), flush=True)
    assert not flag and worst < 2e-5 and eg < 1e-4 and eb < 1e-4


if __name__ == "__main__":
    main()
