"""-m gpu: every HIP kernel, called through the C ABI (sgg_amd.lib.HipKernels), against the kernel-level CPU
reference (oracle/kernels_ref.py, computed in fp64 on the same seeded inputs).

Tolerance: fp32 arithmetic with different summation order than the reference ->
|hip - ref| <= ATOL + RTOL * max|ref| with RTOL = 2e-5 (contractions of up to ~13k terms), stated per test.
Integer outputs (argmax) are compared exactly.
"""
import math

import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=g(seed), dtype=torch.float32) * scale)


def close(hip_t, ref_t, rtol=2e-5, atol=1e-6, what=""):
    h = hip_t.detach().cpu().double()
    r = ref_t.detach().cpu().double()
    assert h.shape == r.shape, (what, h.shape, r.shape)
    assert torch.isfinite(h).all(), what + ": non-finite values"
    tol = atol + rtol * float(r.abs().max())
    err = float((h - r).abs().max())
    assert err <= tol, "%s: max err %.3e > tol %.3e (max|ref| %.3e)" % (what, err, tol, float(r.abs().max()))


def dev(t):
    return t.cuda().contiguous()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride
    (2, 12, 12, 32, 32, 3, 1),
    (2, 12, 10, 32, 64, 3, 1),
    (1, 9, 9, 64, 128, 3, 1),
    (2, 8, 8, 128, 256, 3, 1),
    (1, 6, 6, 256, 512, 3, 1),
    (2, 12, 12, 32, 32, 5, 2),      # even size: SAME pads (1,2)
    (2, 13, 11, 32, 32, 5, 2),      # odd size: pads (2,2)
    (2, 8, 8, 128, 128, 5, 2),
    (1, 4, 4, 512, 512, 5, 2),
    (3, 20, 20, 64, 64, 3, 1),      # M = 1200: several tiles + ragged last tile
    (2, 10, 10, 3, 32, 3, 1),       # conv1_1 path (Cin = 3)
    (2, 16, 16, 32, 32, 5, 2),      # output grid 8x8: wgrad through the halo kernel's four stride-2 parity classes
    (2, 32, 16, 32, 64, 5, 2),
    (3, 16, 32, 64, 128, 5, 2),
    (1, 16, 16, 128, 128, 5, 2),
    # wgrad in row bands (grids that 8x8 blocks do not tile, 64+ channels): 28x28 in 4-row bands, 14x14 in 8-row bands, 12x20 in
    # bands of 5, 5 and 2 rows, 7x7 as one band; the 3x3 cases above with 20x20, 9x9 and 6x6 grids take the same path
    (2, 56, 56, 64, 128, 5, 2),
    (3, 28, 28, 128, 64, 5, 2),
    (2, 24, 40, 64, 64, 5, 2),
    (1, 14, 14, 64, 64, 5, 2),
]


ONE_PIECE_TOL = {1: 2e-3, 4: 1.5e-2}     # single-piece (mixed-precision) modes: operands rounded to fp16 (2^-11) / bf16 (2^-8)


@pytest.fixture(params=[0, 2, 6, 3, 1, 4], ids=["f32mfma", "f16x3", "bf16x6", "bf16x3", "f16x1", "bf16x1"])
def conv_mode(request, hip):
    """conv precision modes of the C ABI: native f32 MFMA, f32 operands scaled and split into two fp16 pieces (3 products),
    or into bf16 pieces (6 / 3 products).  Tolerance vs the fp64 reference: f32, f16x3 and bf16x6 2e-5 * max|ref|
    (f32-equivalent), bf16x3 (drops 2^-17 cross terms) 1e-4; the single-piece modes 1 / 4 (one product of operands rounded to
    fp16 / bf16: mixed-precision arithmetic) 2e-3 / 1.5e-2."""
    old = hip.conv_precision
    hip.conv_precision = request.param
    yield {0: 2e-5, 2: 2e-5, 6: 2e-5, 3: 1e-4, **ONE_PIECE_TOL}[request.param]
    hip.conv_precision = old


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(hip, ref, case, conv_mode):
    B, H, W, Ci, Co, k, s = case
    x = rnd((B, H, W, Ci), 1)
    w = rnd((k, k, Ci, Co), 2, 1.0 / math.sqrt(k * k * Ci))
    b = rnd((Co,), 3, 0.1)
    Ho, Wo = O.same_pads(H, k, s)[0], O.same_pads(W, k, s)[0]
    dy = rnd((B, Ho, Wo, Co), 4)
    # reference in fp64
    xr, wr, br, dyr = x.double(), w.double(), b.double(), dy.double()
    y_ref = torch.empty((B, Ho, Wo, Co), dtype=torch.float64)
    ref.conv_fwd(xr, wr, None, br, y_ref, s)
    dw_ref = torch.empty_like(wr)
    ref.conv_wgrad(xr, dyr, dw_ref, s)
    # hip
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    if Ci == 3:
        wf = wd
    else:
        wf = torch.empty((k, k, Co, Ci), device="cuda")
        hip.hwio_to_hwoi(wd, wf)
        close(wf, w.permute(0, 1, 3, 2), 0, 0, "hwio_to_hwoi")
    y = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    hip.conv_fwd(xd, wd, wf, bd, y, s)
    close(y, y_ref, rtol=conv_mode, what="conv_fwd %s" % (case,))
    if hip.conv_precision and Ci != 3:      # same result with the weights pre-split into bf16 planes
        ws = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda")
        hip.split_weights(wf, ws)
        y2 = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
        hip.conv_fwd(xd, wd, wf, bd, y2, s, ws)
        assert torch.equal(y2, y), "pre-split weights must give bit-identical outputs"
    dw = torch.full_like(wd, float("nan"))
    hip.conv_wgrad(xd, dyd, dw, s)
    close(dw, dw_ref, rtol=conv_mode, what="conv_wgrad %s" % (case,))
    if Ci != 3:
        dx_ref = torch.empty_like(xr)
        ref.conv_dgrad(dyr, wr, dx_ref, s)
        dx = torch.full_like(xd, float("nan"))
        hip.conv_dgrad(dyd, wd, dx, s)
        close(dx, dx_ref, rtol=conv_mode, what="conv_dgrad %s" % (case,))
        if hip.conv_precision:
            ws = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda")
            hip.split_weights(wd, ws)
            dx2 = torch.full_like(xd, float("nan"))
            hip.conv_dgrad(dyd, wd, dx2, s, ws)
            assert torch.equal(dx2, dx)


def test_conv_wgrad_split_k(hip, ref):
    # enough pixels that the pixel range is split over several workgroups (nsplit > 1)
    B, H, W, Ci, Co, k, s = 4, 48, 48, 32, 32, 3, 1
    x, dy = rnd((B, H, W, Ci), 5), rnd((B, H, W, Co), 6)
    dw_ref = torch.empty((k, k, Ci, Co), dtype=torch.float64)
    ref.conv_wgrad(x.double(), dy.double(), dw_ref, s)
    dw = torch.full((k, k, Ci, Co), float("nan"), device="cuda")
    hip.conv_wgrad(dev(x), dev(dy), dw, s)
    close(dw, dw_ref, rtol=5e-5, what="wgrad split-k")


@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (3, 7, 7, 64), (2, 16, 16, 512), (2, 40, 40, 32), (1, 4, 4, 256)])
def test_layernorm_elu(hip, ref, shape):
    B, H, W, C = shape
    y = rnd(shape, 7, 2.0) + 0.3
    gamma, beta = 1.0 + rnd((C,), 8, 0.2), rnd((C,), 9, 0.2)
    da = rnd(shape, 10)
    a_ref = torch.empty(shape, dtype=torch.float64)
    st_ref = torch.empty((B, 2), dtype=torch.float64)
    ref.ln_elu_fwd(y.double(), gamma.double(), beta.double(), a_ref, st_ref)
    dy_ref = torch.empty(shape, dtype=torch.float64)
    dg_ref, db_ref, dbias_ref = (torch.empty(C, dtype=torch.float64) for _ in range(3))
    ref.ln_elu_bwd(y.double(), da.double(), gamma.double(), beta.double(), st_ref, dy_ref, dg_ref, db_ref, dbias_ref)
    yd, gd, bd, dad = dev(y), dev(gamma), dev(beta), dev(da)
    a = torch.full(shape, float("nan"), device="cuda")
    st = torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(yd, gd, bd, a, st)
    close(a, a_ref, what="ln fwd")
    close(st, st_ref, what="ln stats")
    dy = torch.full(shape, float("nan"), device="cuda")
    dg, db, dbias = (torch.full((C,), float("nan"), device="cuda") for _ in range(3))
    hip.ln_elu_bwd(yd, dad, gd, bd, st, dy, dg, db, dbias)
    close(dy, dy_ref, rtol=5e-5, what="ln bwd dy")
    close(dg, dg_ref, rtol=5e-5, what="ln dgamma")
    close(db, db_ref, rtol=5e-5, what="ln dbeta")
    close(dbias, dbias_ref, rtol=5e-5, atol=2e-5, what="ln dbias_prev")


@pytest.mark.parametrize("shape,region", [((2, 16, 16, 32), (1, 1, 15, 15)), ((2, 112, 112, 32), (1, 1, 111, 111)),
                                          ((1, 224, 224, 32), (3, 3, 221, 221)), ((3, 12, 12, 128), (2, 3, 9, 7)),
                                          ((2, 8, 8, 512), (0, 0, 8, 7))])
def test_layernorm_elu_valid_region(hip, ref, shape, region):
    """Canvas mode (odd image sizes, trunk.plan_canvas): statistics / gradients over the window only, zeros written outside;
    canvas pixels outside the window hold garbage (1e6 here) that must not leak into anything."""
    B, H, W, C = shape
    r0, c0, hv, wv = region
    inside = torch.zeros((1, H, W, 1), dtype=torch.bool)
    inside[:, r0:r0 + hv, c0:c0 + wv] = True
    y = torch.where(inside, rnd(shape, 7, 2.0) + 0.3, torch.full(shape, 1e6))
    da = torch.where(inside, rnd(shape, 10), torch.full(shape, -1e6))
    gamma, beta = 1.0 + rnd((C,), 8, 0.2), rnd((C,), 9, 0.2)
    a_ref = torch.empty(shape, dtype=torch.float64)
    st_ref = torch.empty((B, 2), dtype=torch.float64)
    ref.ln_elu_fwd(y.double(), gamma.double(), beta.double(), a_ref, st_ref, region=region)
    dy_ref = torch.empty(shape, dtype=torch.float64)
    dg_ref, db_ref, dbias_ref = (torch.empty(C, dtype=torch.float64) for _ in range(3))
    ref.ln_elu_bwd(y.double(), da.double(), gamma.double(), beta.double(), st_ref, dy_ref, dg_ref, db_ref, dbias_ref, region=region)
    yd, gd, bd, dad = dev(y), dev(gamma), dev(beta), dev(da)
    a = torch.full(shape, float("nan"), device="cuda")
    st = torch.empty((B, 2), device="cuda")
    amax = torch.zeros(2, device="cuda")
    hip.ln_elu_fwd(yd, gd, bd, a, st, amax[0:1], region=region)
    close(a, a_ref, what="ln fwd (region)")
    close(st, st_ref, what="ln stats (region)")
    assert torch.equal(a.cpu() == 0, (a_ref == 0)), "zeros exactly outside the window"
    assert abs(float(amax[0]) - float(a_ref.abs().max())) <= 1e-5 * float(a_ref.abs().max())
    dy = torch.full(shape, float("nan"), device="cuda")
    dg, db, dbias = (torch.full((C,), float("nan"), device="cuda") for _ in range(3))
    hip.ln_elu_bwd(yd, dad, gd, bd, st, dy, dg, db, dbias, amax[1:2], region=region)
    close(dy, dy_ref, rtol=5e-5, what="ln bwd dy (region)")
    close(dg, dg_ref, rtol=5e-5, what="ln dgamma (region)")
    close(db, db_ref, rtol=5e-5, what="ln dbeta (region)")
    close(dbias, dbias_ref, rtol=5e-5, atol=2e-5, what="ln dbias_prev (region)")
    assert abs(float(amax[1]) - float(dy_ref.abs().max())) <= 1e-4 * float(dy_ref.abs().max())


GEMM_CASES = [(8, 16, 8704), (64, 196, 4096), (24, 50, 512), (24, 2048, 562), (64, 300, 1000), (8, 1, 512), (40, 100, 1030), (128, 2048, 1536),
              (70, 130, 33), (192, 2048, 1324)]


@pytest.mark.parametrize("mnk", GEMM_CASES)
def test_gemm_modes(hip, ref, mnk):
    M, N, K = mnk
    A, Bm, bias = rnd((M, K), 11), rnd((K, N), 12), rnd((N,), 13)
    C0 = rnd((M, N), 14)
    # NN (+bias), also with accumulate
    Cr = torch.empty((M, N), dtype=torch.float64)
    ref.gemm_nn(A.double(), Bm.double(), Cr, bias.double())
    C = torch.full((M, N), float("nan"), device="cuda")
    hip.gemm_nn(dev(A), dev(Bm), C, dev(bias))
    close(C, Cr, what="gemm_nn %s" % (mnk,))
    C = dev(C0)
    hip.gemm_nn(dev(A), dev(Bm), C, None, accumulate=True)
    close(C, C0.double() + A.double() @ Bm.double(), what="gemm_nn acc %s" % (mnk,))
    # NT
    Bt = Bm.t().contiguous()
    C = torch.full((M, N), float("nan"), device="cuda")
    hip.gemm_nt(dev(A), dev(Bt), C)
    close(C, A.double() @ Bm.double(), what="gemm_nt %s" % (mnk,))
    # TN
    At = A.t().contiguous()
    C = dev(C0)
    hip.gemm_tn(dev(At), dev(Bm), C, accumulate=True)
    close(C, C0.double() + A.double() @ Bm.double(), what="gemm_tn %s" % (mnk,))


def test_gemm_strided_views(hip):
    # column slices of wider buffers (leading dimension != width), as the heads use them
    R, W0 = 24, 812 + 512
    XH = dev(rnd((R, W0), 15))
    K = dev(rnd((W0, 2048), 16, 0.05))
    G = torch.empty((R, 2048), device="cuda")
    hip.gemm_nn(XH, K, G)
    close(G, XH.cpu().double() @ K.cpu().double(), what="gates gemm")
    out = torch.zeros((R, 3, 50), device="cuda")
    Wd = dev(rnd((512, 50), 17, 0.05))
    bd = dev(rnd((50,), 18))
    hip.gemm_nn(XH[:, 812:], Wd, out[:, 1, :], bd)
    close(out[:, 1, :], XH[:, 812:].cpu().double() @ Wd.cpu().double() + bd.cpu().double(), what="decoder slab")
    assert float(out[:, 0, :].abs().max()) == 0.0 and float(out[:, 2, :].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [(4, 1, 16), (4, 3, 16), (3, 2, 196), (2, 1, 784)])
@pytest.mark.parametrize("dual", [False, True])
def test_attention_step(hip, ref, cfg, dual):
    B, npass, L = cfg
    C, R, np_ = 512, B * npass, (2 if dual else 1)
    P, ctx = rnd((B, L), 20), rnd((B, L, C), 21)
    ec, dz = rnd((np_, R, L), 22), rnd((np_, R, C), 23)
    al_ref = torch.empty((np_, R, L), dtype=torch.float64)
    z_ref = torch.empty((np_, R, C), dtype=torch.float64)
    ref.attn_step_fwd(P.double(), ec.double(), ctx.double(), al_ref, z_ref)
    de_ref = torch.empty((np_, R, L), dtype=torch.float64)
    dP0, dctx0 = rnd((B, L), 24), rnd((B, L, C), 25)
    dP_ref, dctx_ref = dP0.double().clone(), dctx0.double().clone()
    ref.attn_step_bwd(ctx.double(), al_ref, dz.double(), de_ref, dP_ref, dctx_ref, True)
    Pd, ctxd, ecd, dzd = dev(P), dev(ctx), dev(ec), dev(dz)
    al = torch.full((np_, R, L), float("nan"), device="cuda")
    zbuf = torch.full((np_, R, C + 40), float("nan"), device="cuda")     # z lands in a column slice
    hip.attn_step_fwd(Pd, ecd, ctxd, al, zbuf[:, :, :C])
    close(al, al_ref, what="alpha")
    close(zbuf[:, :, :C], z_ref, what="z")
    de = torch.full((np_, R, L), float("nan"), device="cuda")
    dP, dctx = dev(dP0), dev(dctx0)
    hip.attn_step_bwd(ctxd, dev(al_ref.float()), dzd, de, dP, dctx, True)
    close(de, de_ref, rtol=5e-5, what="de")
    close(dP, dP_ref, rtol=5e-5, what="dP")
    close(dctx, dctx_ref, rtol=5e-5, what="dctx")
    # overwrite mode
    dP2, dctx2 = torch.full_like(dP, float("nan")), torch.full_like(dctx, float("nan"))
    hip.attn_step_bwd(ctxd, dev(al_ref.float()), dzd, de, dP2, dctx2, False)
    close(dP2, dP_ref - dP0.double(), rtol=5e-5, atol=1e-5, what="dP overwrite")
    close(dctx2, dctx_ref - dctx0.double(), rtol=5e-5, atol=1e-5, what="dctx overwrite")


@pytest.mark.parametrize("R", [1, 6, 64])
@pytest.mark.parametrize("dual", [False, True])
def test_lnlstm_gates(hip, ref, R, dual):
    np_ = 2 if dual else 1
    gates, c_prev = rnd((np_, R, 2048), 30, 1.5), rnd((np_, R, 512), 31)
    ln = torch.stack([1.0 + rnd((512,), 32 + i, 0.2) if i % 2 == 0 else rnd((512,), 32 + i, 0.2) for i in range(10)])
    dh, dcn = rnd((np_, R, 512), 50), rnd((np_, R, 512), 51)
    cn_ref, h_ref = (torch.empty((np_, R, 512), dtype=torch.float64) for _ in range(2))
    ref.lstm_fwd(gates.double(), c_prev.double(), ln.double(), cn_ref, h_ref)
    dg_ref = torch.empty((np_, R, 2048), dtype=torch.float64)
    dcp_ref = torch.empty((np_, R, 512), dtype=torch.float64)
    pg_ref = torch.empty((R, 10, 512), dtype=torch.float64)
    ref.lstm_bwd(gates.double(), c_prev.double(), ln.double(), dh.double(), dcn.double(), dg_ref, dcp_ref, pg_ref)
    gd, cd, lnd, dhd, dcnd = dev(gates), dev(c_prev), dev(ln), dev(dh), dev(dcn)
    cn = torch.full((np_, R, 512), float("nan"), device="cuda")
    hbuf = torch.full((np_, R, 812 + 512), float("nan"), device="cuda")
    hip.lstm_fwd(gd, cd, lnd, cn, hbuf[:, :, 812:])
    close(cn, cn_ref, what="c_new")
    close(hbuf[:, :, 812:], h_ref, what="h_new")
    dg = torch.full((np_, R, 2048), float("nan"), device="cuda")
    dcp = torch.full((np_, R, 512), float("nan"), device="cuda")
    pg = torch.full((R, 10, 512), float("nan"), device="cuda")
    hip.lstm_bwd(gd, cd, lnd, dhd, dcnd, dg, dcp, pg)
    close(dg, dg_ref, rtol=1e-4, what="dgates")
    close(dcp, dcp_ref, rtol=1e-4, what="dc_prev")
    close(pg, pg_ref, rtol=1e-4, what="pgrad")
    # no cotangent on the new cell state
    ref.lstm_bwd(gates.double(), c_prev.double(), ln.double(), dh.double(), None, dg_ref, dcp_ref, pg_ref)
    hip.lstm_bwd(gd, cd, lnd, dhd, None, dg, dcp, pg)
    close(dg, dg_ref, rtol=1e-4, what="dgates (dc_new=None)")


def test_spatial_mean_colsum(hip, ref):
    B, L, C, npass = 3, 16, 512, 2
    R = B * npass
    ctx = rnd((B, L, C), 60)
    oc = torch.full((R, C), float("nan"), device="cuda")
    ohb = torch.full((R, 300 + C), float("nan"), device="cuda")
    hip.spatial_mean_fwd(dev(ctx), oc, ohb[:, 300:])
    m = ctx.double().mean(dim=1).repeat(npass, 1)
    close(oc, m, what="c0")
    close(ohb[:, 300:], m, what="h0")
    dc0, dh0, dctx0 = rnd((R, C), 61), rnd((R, C), 62), rnd((B, L, C), 63)
    dref = dctx0.double().clone()
    ref.spatial_mean_bwd(dc0.double(), dh0.double(), dref, True)
    d = dev(dctx0)
    hip.spatial_mean_bwd(dev(dc0), dev(dh0), d, True)
    close(d, dref, what="spatial_mean_bwd")
    X = rnd((1337, 5120), 64)
    out = dev(rnd((5120,), 65))
    exp = out.cpu().double() + X.double().sum(0)
    hip.colsum(dev(X), out, True)
    close(out, exp, what="colsum")


def test_loss_adam_argmax(hip, ref):
    B, T, V = 8, 3, 50
    labels = torch.randint(0, V, (B, T), generator=g(70))
    oh = torch.empty((B, T, V), device="cuda")
    hip.onehot(labels.cuda(), oh)
    assert torch.equal(oh.cpu(), torch.nn.functional.one_hot(labels, V).float())
    fake, alpha = rnd((B, T, V), 71), torch.rand((B,), generator=g(72))
    xh = torch.empty((B, T, V), device="cuda")
    hip.interpolate(oh, dev(fake), dev(alpha), xh)
    close(xh, oh.cpu().double() + alpha.double()[:, None, None] * (fake.double() - oh.cpu().double()), what="interpolate")
    gr = rnd((B, T, V), 73, 0.2)
    gr[0] *= 0.01   # a row with slope < 1 (one-sided penalty inactive)
    sl, pen = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    hip.gp_fwd(dev(gr), sl, pen)
    slr, penr = torch.empty(B, dtype=torch.float64), torch.empty(B, dtype=torch.float64)
    ref.gp_fwd(gr.double(), slr, penr)
    close(sl, slr, what="slopes"); close(pen, penr, what="pen")
    assert float(pen[0]) == 0.0
    v = torch.empty((B, T, V), device="cuda")
    hip.gp_bwd(dev(gr), sl, pen, v, 10.0)
    vr = torch.empty((B, T, V), dtype=torch.float64)
    ref.gp_bwd(gr.double(), slr, penr, vr, 10.0)
    close(v, vr, what="gp_bwd")
    d_out = rnd((3 * B, T), 74)
    out4 = torch.empty(4, device="cuda")
    hip.wgan_losses(dev(d_out), pen, 10.0, B, T, True, out4)
    o4 = torch.empty(4, dtype=torch.float64)
    ref.wgan_losses(d_out.double(), penr, 10.0, B, T, True, o4)
    close(out4, o4, what="wgan_losses")
    # TF Adam, 3 steps, odd length (tail path)
    n = 4099
    p, gg = rnd((n,), 75), rnd((n,), 76, 0.1)
    pr, mr, vr_ = p.double().clone(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    buf = torch.zeros((4, 4112), device="cuda")
    pd, gd, md, vd = buf[0, :n], buf[1, :n], buf[2, :n], buf[3, :n]
    pd.copy_(p); gd.copy_(gg)
    for t in range(1, 4):
        lr_t = O.tf_adam_lr_t(t)
        hip.adam(pd, gd, md, vd, lr_t, O.ADAM_B1, O.ADAM_B2, O.ADAM_EPS)
        ref.adam(pr, gg.double(), mr, vr_, lr_t, O.ADAM_B1, O.ADAM_B2, O.ADAM_EPS)
    close(pd, pr, rtol=1e-6, what="adam params")
    close(md, mr, rtol=1e-5, what="adam m")
    # argmax with ties: first index wins
    x = rnd((B * T, V), 77)
    x[0, 7] = x[0, 31] = 100.0
    x[1, :] = 0.0
    out = torch.empty(B * T, dtype=torch.int64, device="cuda")
    hip.argmax_rows(dev(x), out)
    assert torch.equal(out.cpu(), O.argmax_tokens(x))
    assert int(out[0]) == 7 and int(out[1]) == 0


def test_errors_are_loud(hip):
    from sgg_amd.lib import SggError
    x = torch.zeros((1, 4, 4, 24), device="cuda")       # Cin not a multiple of 32
    w = torch.zeros((3, 3, 24, 32), device="cuda")
    y = torch.zeros((1, 4, 4, 32), device="cuda")
    with pytest.raises(SggError):
        hip.conv_fwd(x, w, w, torch.zeros(32, device="cuda"), y, 1)
    with pytest.raises(SggError):
        hip.fill(torch.zeros(4), 1.0)                    # CPU tensor: no fallback


HALO_CASES = [
    # B, H, W, Cin, Cout  (3x3, stride 1, grids divisible by 8)
    (2, 16, 24, 32, 32),
    (1, 8, 8, 32, 64),          # a single 8x8 block: the other blocks of the workgroup are dead
    (3, 8, 16, 64, 64),
    (2, 24, 16, 64, 128),
    (3, 8, 8, 128, 128),        # odd number of blocks
    (1, 16, 16, 128, 256),
    (2, 8, 8, 256, 256),
    (5, 40, 24, 32, 32),        # 75 blocks: several stages per workgroup, ragged last stage
    (3, 24, 40, 64, 64),
    (1, 8, 8, 512, 512),        # 16 channel chunks, 4 n-tiles, one block: most XCD ranges are empty
    (9, 8, 16, 64, 32),         # 18 blocks over 8 XCD ranges of 0..1 tiles (N = 32 tile: 4 blocks per workgroup)
]


@pytest.mark.parametrize("mode", [2, 3, 1, 4], ids=["f16x3", "bf16x3", "f16x1", "bf16x1"])
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_fwd_dgrad(hip, ref, case, mode):
    """Halo-resident 3x3 stride-1 kernel (weights as MFMA fragments) vs the fp64 reference and vs the gather kernel."""
    B, H, W, Ci, Co = case
    old = hip.conv_precision
    hip.conv_precision = mode
    try:
        tol = {2: 2e-5, 3: 1e-4, **ONE_PIECE_TOL}[mode]
        lay_f, lay_b = hip.conv_wsplit_layout(3, 1, H, W, Ci, Co), hip.conv_wsplit_layout(3, 1, H, W, Co, Ci)
        pc_f = mode in (2, 3) and Co % 128 == 0 and Ci % 64 == 0      # producer / consumer kernel (conv_halo_pc.hip): w_split_layout 4
        pc_b = mode in (2, 3) and Ci % 128 == 0 and Co % 64 == 0
        assert lay_f == (4 if pc_f else 1) and lay_b == (4 if pc_b else 1), (lay_f, lay_b)
        assert hip.conv_wsplit_layout(3, 1, H + 1, W, Ci, Co) == 0 and hip.conv_wsplit_layout(5, 2, H, W, Ci, Co) not in (1, 4)
        x, w, b = rnd((B, H, W, Ci), 11), rnd((3, 3, Ci, Co), 12, 1.0 / math.sqrt(9 * Ci)), rnd((Co,), 13, 0.1)
        dy = rnd((B, H, W, Co), 14)
        y_ref = torch.empty((B, H, W, Co), dtype=torch.float64)
        ref.conv_fwd(x.double(), w.double(), None, b.double(), y_ref, 1)
        dx_ref = torch.empty((B, H, W, Ci), dtype=torch.float64)
        ref.conv_dgrad(dy.double(), w.double(), dx_ref, 1)
        xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
        wf = torch.empty((3, 3, Co, Ci), device="cuda")
        hip.hwio_to_hwoi(wd, wf)
        ws_f = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
        ws_b = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
        hip.split_weights(wf, ws_f, layout=1)
        hip.split_weights(wd, ws_b, layout=1)
        y = torch.full((B, H, W, Co), float("nan"), device="cuda")
        hip.conv_fwd(xd, wd, wf, bd, y, 1, ws_f, w_split_layout=1)
        close(y, y_ref, rtol=tol, what="halo conv_fwd %s" % (case,))
        dx = torch.full((B, H, W, Ci), float("nan"), device="cuda")
        hip.conv_dgrad(dyd, wd, dx, 1, ws_b, w_split_layout=1)
        close(dx, dx_ref, rtol=tol, what="halo conv_dgrad %s" % (case,))
        # Conv2DBackpropFilter: the entry point routes 3x3 stride-1 shapes on 8-divisible grids to the halo-resident kernel
        dw_ref = torch.empty((3, 3, Ci, Co), dtype=torch.float64)
        ref.conv_wgrad(x.double(), dy.double(), dw_ref, 1)
        dw = torch.full((3, 3, Ci, Co), float("nan"), device="cuda")
        hip.conv_wgrad(xd, dyd, dw, 1)
        close(dw, dw_ref, rtol=tol, what="halo conv_wgrad %s" % (case,))
        # same operands through the gather kernel: identical pieces and products, only the summation order differs
        y_g = torch.empty_like(y)
        hip.conv_fwd(xd, wd, wf, bd, y_g, 1)
        close(y, y_g.cpu(), rtol=5e-6 if mode in (2, 3) else 2.0 * tol, what="halo vs gather")
        # the producer / consumer kernel (layout 4: fragments of the K = 32 MFMA shape) where the library picks it: same pieces and
        # products again, against fp64 and against the 4-wave kernel
        if lay_f == 4:
            ws4 = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
            hip.split_weights(wf, ws4, layout=4)
            y4 = torch.full((B, H, W, Co), float("nan"), device="cuda")
            hip.conv_fwd(xd, wd, wf, bd, y4, 1, ws4, w_split_layout=4)
            close(y4, y_ref, rtol=tol, what="producer/consumer conv_fwd %s" % (case,))
            close(y4, y.cpu(), rtol=5e-6, what="producer/consumer vs 4-wave kernel")
            nts4 = hip.conv_tile_stats_count((B, H, W, Co), Ci, 3, 1, 4)
            ts4 = torch.full((B, nts4, 4), float("nan"), device="cuda")
            y5 = torch.empty_like(y4)
            hip.conv_fwd(xd, wd, wf, bd, y5, 1, ws4, tile_stats=ts4, w_split_layout=4)
            assert torch.equal(y5, y4)
            g4, b4 = dev(1.0 + rnd((Co,), 15, 0.2)), dev(rnd((Co,), 16, 0.2))
            a_t, a_p = torch.empty_like(y4), torch.empty_like(y4)
            st_t, st_p = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
            hip.ln_elu_fwd(y4, g4, b4, a_t, st_t, tile_stats=ts4)
            hip.ln_elu_fwd(y4, g4, b4, a_p, st_p)
            close(st_t, st_p.cpu(), rtol=1e-6, what="producer/consumer tile statistics vs statistics pass")
        if lay_b == 4:
            ws4b = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
            hip.split_weights(wd, ws4b, layout=4)
            dx4 = torch.full((B, H, W, Ci), float("nan"), device="cuda")
            hip.conv_dgrad(dyd, wd, dx4, 1, ws4b, w_split_layout=4)
            close(dx4, dx_ref, rtol=tol, what="producer/consumer conv_dgrad %s" % (case,))
            close(dx4, dx.cpu(), rtol=5e-6, what="producer/consumer dgrad vs 4-wave kernel")
        # LayerNorm partial statistics from the halo epilogue
        nts = hip.conv_tile_stats_count((B, H, W, Co), Ci, 3, 1, 1)
        assert nts in ((H * W // 64) * (Co // (64 if Co % 64 == 0 else 32)),
                       (H * W // 64) * (Co // (64 if Co % 128 == 0 else 32)),        # (per 64 columns on the 128-column tiling, per 32 otherwise; first form: -DSGG_HALO_N64_NB2=0 builds)
                       (H * W // 64) * (Co // 32))                                  # (-DSGG_HALO_N128_WB2=1 builds: a wave owns two blocks x 32 columns)
        ts = torch.full((B, nts, 4), float("nan"), device="cuda")
        y2 = torch.empty_like(y)
        hip.conv_fwd(xd, wd, wf, bd, y2, 1, ws_f, tile_stats=ts, w_split_layout=1)
        assert torch.equal(y2, y)
        gamma, beta = dev(1.0 + rnd((Co,), 15, 0.2)), dev(rnd((Co,), 16, 0.2))
        a1, a2 = torch.empty_like(y), torch.empty_like(y)
        st1, st2 = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
        hip.ln_elu_fwd(y, gamma, beta, a1, st1, tile_stats=ts)
        hip.ln_elu_fwd(y, gamma, beta, a2, st2)
        close(st1, st2.cpu(), rtol=1e-6, what="stats from the halo epilogue vs statistics pass")
        close(a1, a2.cpu(), rtol=1e-6, what="LN output")
    finally:
        hip.conv_precision = old


@pytest.mark.parametrize("shape", [(2, 16, 32), (3, 13, 45), (1, 8, 64)])
def test_conv_c3_tile_stats(hip, ref, shape):
    """conv1_1 (Cin = 3): output vs fp64 and the per-tile LayerNorm partials (ragged tiles at the right / bottom edge)."""
    B, H, W = shape
    x, w, b = rnd((B, H, W, 3), 70), rnd((3, 3, 3, 32), 71, 0.2), rnd((32,), 72, 0.5)
    xd, wd, bd = dev(x), dev(w), dev(b)
    nts = hip.conv_tile_stats_count((B, H, W, 32), 3, 3, 1, 0)
    assert nts == 4 * (-(-H // 8) * -(-W // 32))          # one record per wave: two rows x 32 columns of an 8 x 32 tile
    ts = torch.full((B, nts, 4), float("nan"), device="cuda")
    y = torch.full((B, H, W, 32), float("nan"), device="cuda")
    hip.conv_fwd(xd, wd, wd, bd, y, 1, tile_stats=ts)
    y_ref = torch.empty((B, H, W, 32), dtype=torch.float64)
    ref.conv_fwd(x.double(), w.double(), None, b.double(), y_ref, 1)
    close(y, y_ref, rtol=2e-6, what="conv1_1 forward")
    assert abs(float(ts[:, :, 0].sum()) - B * H * W * 32) < 0.5
    gamma, beta = dev(1.0 + rnd((32,), 73, 0.2)), dev(rnd((32,), 74, 0.2))
    a1, a2 = torch.empty_like(y), torch.empty_like(y)
    st1, st2 = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(y, gamma, beta, a1, st1, tile_stats=ts)
    hip.ln_elu_fwd(y, gamma, beta, a2, st2)
    close(st1, st2.cpu(), rtol=1e-6, what="stats from the conv1_1 epilogue vs statistics pass")
    close(a1, a2.cpu(), rtol=1e-6, what="LN output")


def test_conv_halo_rejects_bad_shapes(hip):
    from sgg_amd.lib import SggError
    if hip.conv_precision not in (2, 3):
        pytest.skip("halo kernel exists for precision 2 / 3")
    x = torch.zeros((1, 12, 12, 32), device="cuda")      # 12 % 8 != 0
    w = torch.zeros((3, 3, 32, 32), device="cuda")
    ws = torch.zeros((2, w.numel()), dtype=torch.int16, device="cuda")
    with pytest.raises(SggError):
        hip.conv_fwd(x, w, w, torch.zeros(32, device="cuda"), torch.zeros_like(x), 1, ws, w_split_layout=1)


@pytest.mark.parametrize("cout", [32, 64, 128])
def test_conv_epilogue_layernorm_stats(hip, ref, cout):
    """The split conv kernels can emit per-tile (count, mean, M2) of their output; LayerNorm merged from those
    must equal LayerNorm with its own statistics pass (and the fp64 reference)."""
    if hip.conv_precision == 0:
        pytest.skip("native f32 kernels do not emit tile statistics")
    B, H, W, Ci, k, s = 3, 16, 16, 64, 3, 1
    x, w, b = rnd((B, H, W, Ci), 90), rnd((k, k, Ci, cout), 91, 0.05), rnd((cout,), 92, 0.5)
    gamma, beta = 1.0 + rnd((cout,), 93, 0.2), rnd((cout,), 94, 0.2)
    xd, wd, bd = dev(x), dev(w), dev(b)
    wf = torch.empty((k, k, cout, Ci), device="cuda")
    hip.hwio_to_hwoi(wd, wf)
    nts = hip.conv_tile_stats_count((B, H, W, cout), Ci)
    assert nts > 0
    ts = torch.full((B, nts, 4), float("nan"), device="cuda")
    y = torch.empty((B, H, W, cout), device="cuda")
    hip.conv_fwd(xd, wd, wf, bd, y, s, tile_stats=ts)
    a1, a2 = torch.empty_like(y), torch.empty_like(y)
    st1, st2 = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(y, dev(gamma), dev(beta), a1, st1, tile_stats=ts)
    hip.ln_elu_fwd(y, dev(gamma), dev(beta), a2, st2)
    close(st1, st2.cpu(), rtol=1e-6, what="stats from conv epilogue vs statistics pass")
    close(a1, a2.cpu(), rtol=1e-6, what="LN output")
    y_ref = torch.empty((B, H, W, cout), dtype=torch.float64)
    ref.conv_fwd(x.double(), w.double(), None, b.double(), y_ref, s)
    a_ref, st_ref = torch.empty_like(y_ref), torch.empty((B, 2), dtype=torch.float64)
    ref.ln_elu_fwd(y_ref, gamma.double(), beta.double(), a_ref, st_ref)
    close(a1, a_ref, rtol=5e-5, what="conv + LN vs fp64")


S2_CASES = [
    # B, H (= W, even), Cin, Cout   (5x5, stride 2: the half-resolution grid is H/2 x H/2, cut into flat bands of 224 positions)
    (2, 16, 32, 128),       # 128 positions: one ragged band holding two images
    (3, 28, 64, 128),       # 14x14 grid: bands cross image boundaries (2.6 bands)
    (2, 56, 128, 256),      # 28x28 grid: band 3 takes rows 24..27 of image 0 and 0..3 of image 1; two n-tiles
    (1, 112, 32, 128),      # 56x56 grid: 14 aligned bands of four rows (LayerNorm partials available)
    (2, 24, 32, 128),       # 12x12 grid: bands start in the middle of a row
    (5, 28, 256, 512),      # 980 positions, ragged last band, four n-tiles, 16 chunks
    (70, 4, 32, 128),       # 2x2 grid: a band spans up to 56 images (224 patch rows of two slots)
    (20, 8, 32, 128),       # 4x4 grid: a band of 128 positions spans 8 images (each with its own zero rows in the padded row space)
    (24, 112, 32, 128),     # 336 bands of 224 positions: more work items than CUs -> the 7-tile variant (the small cases above run
                            # the 4-tile variant unless LayerNorm partials are requested)
    # round 4: more shapes whose bands cross image boundaries mid-row (added with the balanced-band experiment)
    (12, 28, 64, 128),      # 14x14 grid: 10.5 bands, every band crosses an image boundary mid-row
    (9, 56, 64, 256),       # 28x28 grid: 31.5 bands (ragged last band and row tile), two n-tiles
    (160, 8, 32, 128),      # 4x4 grid: a band spans fourteen images
]


@pytest.mark.parametrize("mode", [2, 3, 1, 4], ids=["f16x3", "bf16x3", "f16x1", "bf16x1"])
@pytest.mark.parametrize("case", S2_CASES)
def test_conv_s2_fwd_dgrad(hip, ref, case, mode):
    """Band-resident 5x5 stride-2 kernel (csrc/conv_s2.hip) vs the fp64 reference and vs the gather kernel."""
    from tests import conv_ref64 as R64
    B, H, Ci, Co = case
    old = hip.conv_precision
    hip.conv_precision = mode
    try:
        tol = {2: 2e-5, 3: 1e-4, **ONE_PIECE_TOL}[mode]
        vs_gather = 5e-6 if mode in (2, 3) else 2.0 * tol     # (the gather kernel runs the single-piece modes as two-piece ones)
        assert hip.conv_wsplit_layout(5, 2, H, H, Ci, Co) == 2
        assert hip.conv_wsplit_layout(5, 2, H + 1, H, Ci, Co) == 0          # odd sizes have different SAME pads: gather kernel
        x, w, b = rnd((B, H, H, Ci), 11), rnd((5, 5, Ci, Co), 12, 1.0 / math.sqrt(25 * Ci)), rnd((Co,), 13, 0.1)
        Ho = H // 2
        dy = rnd((B, Ho, Ho, Co), 14)
        y_ref = R64.conv_fwd64(x.double(), w.double(), b.double(), 2)
        xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
        wf = torch.empty((5, 5, Co, Ci), device="cuda")
        hip.hwio_to_hwoi(wd, wf)
        ws_f = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
        hip.split_weights(wf, ws_f, layout=2)
        y = torch.full((B, Ho, Ho, Co), float("nan"), device="cuda")
        hip.conv_fwd(xd, wd, wf, bd, y, 2, ws_f, w_split_layout=2)
        close(y, y_ref, rtol=tol, what="s2 conv_fwd %s" % (case,))
        y_g = torch.empty_like(y)
        hip.conv_fwd(xd, wd, wf, bd, y_g, 2)
        close(y, y_g.cpu(), rtol=vs_gather, what="s2 vs gather (forward)")
        if hip.conv_wsplit_layout(5, 2, H, H, Co, Ci) == 2:                  # dgrad direction: output channels = Cin
            dx_ref = R64.conv_dgrad64(dy.double(), w.double(), (H, H), 2)
            ws_b = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
            hip.split_weights(wd, ws_b, layout=2)
            dx = torch.full((B, H, H, Ci), float("nan"), device="cuda")
            hip.conv_dgrad(dyd, wd, dx, 2, ws_b, w_split_layout=2)
            close(dx, dx_ref, rtol=tol, what="s2 conv_dgrad %s" % (case,))
            dx_g = torch.empty_like(dx)
            hip.conv_dgrad(dyd, wd, dx_g, 2)
            close(dx, dx_g.cpu(), rtol=vs_gather, what="s2 vs gather (dgrad)")
        else:
            assert Ci % 128 != 0
        nts = hip.conv_tile_stats_count((B, Ho, Ho, Co), Ci, 5, 2, 2)
        assert nts == ((Ho * Ho // 224) * (Co // 32) if (Ho * Ho) % 224 == 0 else 0)
        if nts:
            ts = torch.full((B, nts, 4), float("nan"), device="cuda")
            y2 = torch.empty_like(y)
            hip.conv_fwd(xd, wd, wf, bd, y2, 2, ws_f, tile_stats=ts, w_split_layout=2)
            assert torch.equal(y2, y)
            gamma, beta = dev(1.0 + rnd((Co,), 15, 0.2)), dev(rnd((Co,), 16, 0.2))
            a1, a2 = torch.empty_like(y), torch.empty_like(y)
            st1, st2 = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
            hip.ln_elu_fwd(y, gamma, beta, a1, st1, tile_stats=ts)
            hip.ln_elu_fwd(y, gamma, beta, a2, st2)
            close(st1, st2.cpu(), rtol=1e-6, what="stats from the s2 epilogue vs statistics pass")
            close(a1, a2.cpu(), rtol=1e-6, what="LN output")
        if mode in (2, 3):
            # LN prologue of the band-resident forward: x is the producing layer's pre-LayerNorm output; bands that cross image
            # boundaries take each item's (mean, rstd) from its own sample
            assert hip.ln_prologue_fwd_ok(5, 2, H, H, Ci, Co)
            y0 = rnd((B, H, H, Ci), 31, 2.0) + 0.7
            y0[0] *= 3.0
            g2, b2 = dev(1.0 + rnd((Ci,), 32, 0.3)), dev(rnd((Ci,), 33, 0.3))
            y0d = dev(y0)
            a = torch.empty((B, H, H, Ci), device="cuda")
            st = torch.empty((B, 2), device="cuda")
            am = torch.zeros(2, device="cuda")
            hip.ln_elu_fwd(y0d, g2, b2, a, st, am[0:1])
            hip.absmax(wf, am[1:2])
            y_u = torch.empty_like(y)
            hip.conv_fwd(a, wd, wf, bd, y_u, 2, ws_f, am[0:1], am[1:2], None, 2)
            y_f = torch.full(tuple(y.shape), float("nan"), device="cuda")
            hip.conv_fwd(y0d, wd, wf, bd, y_f, 2, ws_f, am[0:1], am[1:2], None, 2, ln=(st, g2, b2))
            close(y_f, y_u.cpu(), rtol=5e-6, what="s2 forward: LN prologue vs unfused %s" % (case,))
    finally:
        hip.conv_precision = old


S2D_CASES = [(2, 16, 16), (3, 48, 32), (1, 64, 112), (5, 32, 16)]      # B, H, W (32 -> 32 channels, H % 16 == W % 16 == 0)


@pytest.mark.parametrize("mode", [2, 3, 1], ids=["f16x3", "bf16x3", "f16x1"])
@pytest.mark.parametrize("case", S2D_CASES)
def test_conv_s2d_fwd_dgrad(hip, ref, case, mode):
    """conv1_3 (5x5 stride 2, 32 -> 32 channels) on the halo-resident kernel as a 3x3 convolution over the space-to-depth view of
    its input (w_split_layout 3): forward (+ LayerNorm partials) and dgrad vs the fp64 reference and vs the gather kernel; the 9-tap
    kernel of sgg_conv_s2d_weights vs its definition."""
    from tests import conv_ref64 as R64
    B, H, W = case
    Ci = Co = 32
    old = hip.conv_precision
    hip.conv_precision = mode
    try:
        tol = {2: 2e-5, 3: 1e-4, **ONE_PIECE_TOL}[mode]
        vs_gather = 5e-6 if mode in (2, 3) else 2.0 * tol
        assert hip.conv_wsplit_layout(5, 2, H, W, Ci, Co) == 3
        assert hip.conv_wsplit_layout(5, 2, H + 8, W, Ci, Co) == 0 and hip.conv_wsplit_layout(5, 2, H, W, Ci, 64) == 0
        x, w, b = rnd((B, H, W, Ci), 11), rnd((5, 5, Ci, Co), 12, 1.0 / math.sqrt(25 * Ci)), rnd((Co,), 13, 0.1)
        Ho, Wo = H // 2, W // 2
        dy = rnd((B, Ho, Wo, Co), 14)
        xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
        w3 = torch.full((3, 3, 4 * Ci, Co), float("nan"), device="cuda")
        hip.s2d_weights(wd, w3)
        w3_ref = torch.zeros((3, 3, 2, 2, Ci, Co))
        for u in range(3):
            for v in range(3):
                for qy in range(2):
                    for qx in range(2):
                        kh, kw = 2 * u + qy - 1, 2 * v + qx - 1
                        if 0 <= kh < 5 and 0 <= kw < 5:
                            w3_ref[u, v, qy, qx] = w[kh, kw]
        assert torch.equal(w3.cpu(), w3_ref.reshape(3, 3, 4 * Ci, Co))
        w3f = torch.empty((3, 3, Co, 4 * Ci), device="cuda")
        hip.hwio_to_hwoi(w3, w3f)
        wf = torch.empty((5, 5, Co, Ci), device="cuda")
        hip.hwio_to_hwoi(wd, wf)
        ws_f = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda")
        ws_b = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda")
        hip.split_weights(w3f, ws_f, layout=3)
        hip.split_weights(w3, ws_b, layout=3)
        y_ref = R64.conv_fwd64(x.double(), w.double(), b.double(), 2)
        y = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
        hip.conv_fwd(xd, wd, wf, bd, y, 2, ws_f, w_split_layout=3)
        close(y, y_ref, rtol=tol, what="s2d conv_fwd %s" % (case,))
        y_g = torch.empty_like(y)
        hip.conv_fwd(xd, wd, wf, bd, y_g, 2)
        close(y, y_g.cpu(), rtol=vs_gather, what="s2d vs gather (forward)")
        dx_ref = R64.conv_dgrad64(dy.double(), w.double(), (H, W), 2)
        dx = torch.full((B, H, W, Ci), float("nan"), device="cuda")
        hip.conv_dgrad(dyd, wd, dx, 2, ws_b, w_split_layout=3)
        close(dx, dx_ref, rtol=tol, what="s2d conv_dgrad %s" % (case,))
        dx_g = torch.empty_like(dx)
        hip.conv_dgrad(dyd, wd, dx_g, 2)
        close(dx, dx_g.cpu(), rtol=vs_gather, what="s2d vs gather (dgrad)")
        nts = hip.conv_tile_stats_count((B, Ho, Wo, Co), Ci, 5, 2, 3)
        assert nts == Ho * Wo // 64
        ts = torch.full((B, nts, 4), float("nan"), device="cuda")
        y2 = torch.empty_like(y)
        hip.conv_fwd(xd, wd, wf, bd, y2, 2, ws_f, tile_stats=ts, w_split_layout=3)
        assert torch.equal(y2, y)
        gamma, beta = dev(1.0 + rnd((Co,), 15, 0.2)), dev(rnd((Co,), 16, 0.2))
        a1, a2 = torch.empty_like(y), torch.empty_like(y)
        st1, st2 = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
        hip.ln_elu_fwd(y, gamma, beta, a1, st1, tile_stats=ts)
        hip.ln_elu_fwd(y, gamma, beta, a2, st2)
        close(st1, st2.cpu(), rtol=1e-6, what="stats from the s2d epilogue vs statistics pass")
        close(a1, a2.cpu(), rtol=1e-6, what="LN output")
        if mode in (2, 3):
            # LN prologue through the space-to-depth view (forward-only passes): x is the producing layer's pre-LayerNorm output;
            # the four 32-channel chunks of the view are the SAME 32 channels (gamma / beta index = channel within the chunk)
            assert hip.ln_prologue_fwd_ok(5, 2, H, W, Ci, Co) and hip.ln_prologue_ok(5, 2, H, W, Ci, Co)
            y0 = rnd((B, H, W, Ci), 31, 2.0) + 0.7
            y0[0] *= 3.0
            g2, b2 = dev(1.0 + rnd((Ci,), 32, 0.3)), dev(rnd((Ci,), 33, 0.3))
            y0d = dev(y0)
            a = torch.empty((B, H, W, Ci), device="cuda")
            st = torch.empty((B, 2), device="cuda")
            am = torch.zeros(2, device="cuda")
            hip.ln_elu_fwd(y0d, g2, b2, a, st, am[0:1])
            hip.absmax(w3f, am[1:2])
            y_u = torch.empty_like(y)
            hip.conv_fwd(a, wd, wf, bd, y_u, 2, ws_f, am[0:1], am[1:2], None, 3)
            y_f = torch.full(tuple(y.shape), float("nan"), device="cuda")
            hip.conv_fwd(y0d, wd, wf, bd, y_f, 2, ws_f, am[0:1], am[1:2], None, 3, ln=(st, g2, b2))
            close(y_f, y_u.cpu(), rtol=5e-6, what="s2d forward: LN prologue vs unfused")
            # ... and the filter gradient (four parity-class launches of the halo-resident wgrad kernel, each with the prologue)
            amdy = torch.zeros(1, device="cuda")
            hip.absmax(dyd, amdy)
            dw_u = torch.full((5, 5, Ci, Co), float("nan"), device="cuda")
            hip.conv_wgrad(a, dyd, dw_u, 2, am[0:1], amdy)
            dw_f = torch.full((5, 5, Ci, Co), float("nan"), device="cuda")
            hip.conv_wgrad(y0d, dyd, dw_f, 2, am[0:1], amdy, ln=(st, g2, b2))
            close(dw_f, dw_u.cpu(), rtol=5e-6, what="conv1_3 wgrad: LN prologue vs unfused")
    finally:
        hip.conv_precision = old


LNP_CASES = [(2, 16, 24, 32, 32), (3, 8, 16, 64, 64), (2, 24, 16, 64, 128), (3, 8, 8, 128, 128), (1, 16, 16, 128, 256), (2, 8, 8, 256, 256),
             (2, 16, 16, 32, 64), (1, 8, 8, 512, 512)]


@pytest.mark.parametrize("mode", [2, 3], ids=["f16x3", "bf16x3"])
@pytest.mark.parametrize("case", LNP_CASES)
def test_layernorm_prologue_fwd_wgrad(hip, ref, case, mode):
    """LN prologue: the halo-resident forward / wgrad kernels fed with the producing layer's PRE-LayerNorm output y apply
    ELU(LN(y)) while staging their patches (generator_with_attention.py:30..56) - against conv(ELU(LN(y))) in fp64 and against the
    unfused HIP path; the statistics come from sgg_layernorm_hwc_finalize over tile partials."""
    from tests import conv_ref64 as R64
    B, H, W, Ci, Co = case
    old = hip.conv_precision
    hip.conv_precision = mode
    try:
        tol = {2: 2e-5, 3: 1e-4}[mode]
        assert hip.ln_prologue_ok(3, 1, H, W, Ci, Co)
        y0 = rnd((B, H, W, Ci), 31, 2.0) + 0.7
        y0[0] *= 3.0                                       # samples with different statistics
        gamma, beta = 1.0 + rnd((Ci,), 32, 0.3), rnd((Ci,), 33, 0.3)
        w, b = rnd((3, 3, Ci, Co), 34, 1.0 / math.sqrt(9 * Ci)), rnd((Co,), 35, 0.1)
        dy = rnd((B, H, W, Co), 36)
        a_ref = torch.empty((B, H, W, Ci), dtype=torch.float64)
        st_ref = torch.empty((B, 2), dtype=torch.float64)
        ref.ln_elu_fwd(y0.double(), gamma.double(), beta.double(), a_ref, st_ref)
        y_ref = R64.conv_fwd64(a_ref, w.double(), b.double(), 1)
        dw_ref = R64.conv_wgrad64(a_ref, dy.double(), 3, 1)
        y0d, gd, bd2, wd, bd, dyd = dev(y0), dev(gamma), dev(beta), dev(w), dev(b), dev(dy)
        # tile partials (count, mean, M2, max |y - mean|) in the [B, n, 4] format the conv epilogues emit, four tiles per sample
        nt = 4
        flat = y0.reshape(B, nt, -1).double()
        ts = torch.empty((B, nt, 4), dtype=torch.float64)
        ts[:, :, 0] = flat.shape[2]
        ts[:, :, 1] = flat.mean(dim=2)
        ts[:, :, 2] = ((flat - flat.mean(dim=2, keepdim=True)) ** 2).sum(dim=2)
        ts[:, :, 3] = (flat - flat.mean(dim=2, keepdim=True)).abs().amax(dim=2)
        tsd = dev(ts.float())
        stats = torch.full((B, 2), float("nan"), device="cuda")
        am = torch.zeros(3, device="cuda")
        hip.ln_finalize(tsd, gd, bd2, stats, am[0:1] if mode == 2 else None, H * W)
        close(stats, st_ref, rtol=1e-5, what="finalize stats")
        if mode == 2:
            assert float(am[0]) >= float(a_ref.abs().max()) * (1 - 1e-6), "published amax must bound max|a|"
            assert float(am[0]) <= 64 * float(a_ref.abs().max()) + 1.0, "bound uselessly loose"
        hip.absmax(wd, am[1:2])
        hip.absmax(dyd, am[2:3])
        wf = torch.empty((3, 3, Co, Ci), device="cuda")
        hip.hwio_to_hwoi(wd, wf)
        ws_f = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
        hip.split_weights(wf, ws_f, am[1:2], layout=1)
        ln = (stats, gd, bd2)
        y = torch.full((B, H, W, Co), float("nan"), device="cuda")
        hip.conv_fwd(y0d, wd, wf, bd, y, 1, ws_f, am[0:1], am[1:2], None, 1, ln=ln)
        close(y, y_ref, rtol=tol, what="forward with LN prologue %s" % (case,))
        dw = torch.full((3, 3, Ci, Co), float("nan"), device="cuda")
        hip.conv_wgrad(y0d, dyd, dw, 1, am[0:1], am[2:3], ln=ln)
        close(dw, dw_ref, rtol=tol, what="wgrad with LN prologue %s" % (case,))
        # the unfused HIP path on the same data
        a = torch.empty((B, H, W, Ci), device="cuda")
        st2 = torch.empty((B, 2), device="cuda")
        hip.ln_elu_fwd(y0d, gd, bd2, a, st2)
        y_u = torch.empty_like(y)
        hip.conv_fwd(a, wd, wf, bd, y_u, 1, ws_f, w_split_layout=1)
        close(y, y_u.cpu(), rtol=5e-6, what="fused vs unfused forward")
        if hip.conv_wsplit_layout(3, 1, H, W, Ci, Co) == 4:      # the producer / consumer kernel applies the prologue in its patch waves
            ws4 = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
            hip.split_weights(wf, ws4, am[1:2], layout=4)
            y4 = torch.full((B, H, W, Co), float("nan"), device="cuda")
            hip.conv_fwd(y0d, wd, wf, bd, y4, 1, ws4, am[0:1], am[1:2], None, 4, ln=ln)
            close(y4, y_ref, rtol=tol, what="producer/consumer forward with LN prologue %s" % (case,))
            close(y4, y.cpu(), rtol=5e-6, what="producer/consumer vs 4-wave kernel, LN prologue")
        dw_u = torch.empty_like(dw)
        hip.conv_wgrad(a, dyd, dw_u, 1)
        close(dw, dw_u.cpu(), rtol=5e-6, what="fused vs unfused wgrad")
    finally:
        hip.conv_precision = old


@pytest.mark.parametrize("mode", [2, 3, 6, 0, 1], ids=["f16x3", "bf16x3", "bf16x6", "f32mfma", "f16x1"])
def test_prepare_weights_equals_per_layer_entry_points(hip, mode):
    """sgg_conv_prepare_weights (three launches for a whole encoder) writes bit for bit what sgg_hwio_to_hwoi + sgg_absmax +
    sgg_conv_s2d_weights + sgg_conv_split_weights(_frag) write layer by layer, in every operand layout."""
    old = hip.conv_precision
    hip.conv_precision = mode
    try:
        # (k, cin, cout, layout_fwd, layout_bwd): planes, halo fragments, band fragments (25 taps), space-to-depth, mixed, ragged tiles
        specs = [(3, 32, 32, 1, 1), (3, 64, 128, 1, 1), (5, 128, 128, 2, 2), (5, 32, 32, 3, 3), (3, 32, 64, 0, 0), (5, 256, 512, 2, 0), (3, 96, 40, 0, 0)]
        if mode in (6, 0):
            specs = [(k, ci, co, 0, 0) for (k, ci, co, _, _) in specs]
        lays, refs = [], []
        amax = torch.full((16,), 7.0, device="cuda")           # (stale values: the call must reset the words it owns)
        amax_ref = torch.zeros(16, device="cuda")
        for j, (k, ci, co, lf, lb) in enumerate(specs):
            w = dev(rnd((k, k, ci, co), 100 + j, 0.1 * (j + 1)))
            s2d = 3 in (lf, lb)
            mk = lambda *shape: torch.full(shape, float("nan"), device="cuda")
            lay = {"w": w, "w_fwd": mk(k, k, co, ci), "w3": mk(3, 3, 4 * ci, co) if s2d else None, "w3_fwd": mk(3, 3, co, 4 * ci) if s2d else None,
                   "ws_fwd": torch.zeros((3, w.numel()), dtype=torch.int16, device="cuda") if mode else None,
                   "ws_bwd": torch.zeros((3, w.numel()), dtype=torch.int16, device="cuda") if mode else None,
                   "amax": amax[j:j + 1] if mode in (1, 2) else None, "ws_layout": lf, "ws_layout_bwd": lb}
            lays.append(lay)
            # the per-layer entry points
            r = {"w_fwd": mk(k, k, co, ci)}
            hip.hwio_to_hwoi(w, r["w_fwd"])
            am = amax_ref[j:j + 1] if mode in (1, 2) else None
            if am is not None:
                hip.absmax(w, am)
            if s2d:
                r["w3"], r["w3_fwd"] = mk(3, 3, 4 * ci, co), mk(3, 3, co, 4 * ci)
                hip.s2d_weights(w, r["w3"])
                hip.hwio_to_hwoi(r["w3"], r["w3_fwd"])
            if mode:
                r["ws_fwd"], r["ws_bwd"] = torch.zeros_like(lay["ws_fwd"]), torch.zeros_like(lay["ws_bwd"])
                hip.split_weights(r["w3_fwd"] if lf == 3 else r["w_fwd"], r["ws_fwd"], am, lf)
                hip.split_weights(r["w3"] if lb == 3 else w, r["ws_bwd"], am, lb)
            refs.append(r)
        descs = hip.weight_descs(lays)
        hip.prepare_weights(descs)
        torch.cuda.synchronize()
        for j, (lay, r) in enumerate(zip(lays, refs)):
            for key, t in r.items():
                assert torch.equal(lay[key], t), "layer %d (%s): %s differs" % (j, specs[j], key)
        if mode in (1, 2):
            assert torch.equal(amax[:len(specs)], amax_ref[:len(specs)]) and float(amax[len(specs)]) == 7.0
    finally:
        hip.conv_precision = old


def test_layernorm_bwd_deferred_finalize_equals_immediate(hip):
    """sgg_layernorm_hwc_elu_bwd with dgamma = NULL leaves the partial sums in the layer's workspace; one
    sgg_layernorm_hwc_bwd_finalize over several layers (one with a valid window) writes bit for bit what the immediate path writes."""
    shapes = [((3, 16, 16, 32), None), ((2, 8, 8, 512), None), ((4, 12, 12, 128), (2, 1, 9, 10)), ((2, 40, 40, 64), None)]
    lays, refs = [], []
    for j, (shape, region) in enumerate(shapes):
        B, H, W, C = shape
        y, da = dev(rnd(shape, 70 + j, 2.0) + 0.3), dev(rnd(shape, 80 + j))
        gamma, beta = dev(1.0 + rnd((C,), 90 + j, 0.2)), dev(rnd((C,), 95 + j, 0.2))
        a, st = torch.empty(shape, device="cuda"), torch.empty((B, 2), device="cuda")
        hip.ln_elu_fwd(y, gamma, beta, a, st, region=region)
        mk = lambda: torch.full((C,), float("nan"), device="cuda")
        dy1, dg1, db1, dbias1 = torch.empty(shape, device="cuda"), mk(), mk(), mk()
        hip.ln_elu_bwd(y, da, gamma, beta, st, dy1, dg1, db1, dbias1, region=region)
        ws = torch.empty(hip.ln_workspace_bytes(shape), dtype=torch.uint8, device="cuda")
        dy2, dg2, db2, dbias2 = torch.empty(shape, device="cuda"), mk(), mk(), (mk() if j != 1 else None)
        hip.ln_elu_bwd(y, da, gamma, beta, st, dy2, None, None, None, region=region, ws=ws)
        assert torch.equal(dy1, dy2)
        lays.append({"ws": ws, "gamma": gamma, "stats": st, "dgamma": dg2, "dbeta": db2, "dbias": dbias2, "shape": shape, "region": region})
        refs.append((dg1, db1, dbias1))
    hip.ln_bwd_finalize(hip.ln_finalize_descs(lays))
    torch.cuda.synchronize()
    for lay, (dg1, db1, dbias1) in zip(lays, refs):
        assert torch.equal(lay["dgamma"], dg1) and torch.equal(lay["dbeta"], db1)
        assert lay["dbias"] is None or torch.equal(lay["dbias"], dbias1)


@pytest.mark.parametrize("shape", [(2, 16, 32), (3, 13, 45), (1, 8, 64), (4, 64, 64)])
def test_conv1_1_filter_gradient_fused_with_layernorm_backward(hip, ref, shape):
    """sgg_layernorm_hwc_elu_bwd_sums + sgg_conv2d_nhwc_wgrad_c3_ln: conv1_1's filter gradient with dy = LayerNormBackward(y, da)
    computed inside the kernel (generator_with_attention.py:29-30 under optimizer.minimize) against fp64 autograd through
    ELU(LN(y)) -> Conv2DBackpropFilter, against the unfused HIP path (apply pass, then the plain conv1_1 filter gradient: same f32
    arithmetic up to the contraction of the compiler), and the deferred parameter-gradient reductions bit for bit (the partial sums in
    the workspace are those of sgg_layernorm_hwc_elu_bwd).  Ragged tiles at the right / bottom edge included."""
    B, H, W = shape
    x, y, da = rnd((B, H, W, 3), 1), rnd((B, H, W, 32), 2, 1.7) + 0.3, rnd((B, H, W, 32), 3)
    gamma, beta = 1.0 + rnd((32,), 4, 0.2), rnd((32,), 5, 0.2)
    # fp64 reference
    dy_ref, dg_ref, db_ref = torch.empty((B, H, W, 32), dtype=torch.float64), torch.empty(32, dtype=torch.float64), torch.empty(32, dtype=torch.float64)
    st64 = torch.empty((B, 2), dtype=torch.float64)
    a64 = torch.empty((B, H, W, 32), dtype=torch.float64)
    ref.ln_elu_fwd(y.double(), gamma.double(), beta.double(), a64, st64)
    ref.ln_elu_bwd(y.double(), da.double(), gamma.double(), beta.double(), st64, dy_ref, dg_ref, db_ref, None)
    dw_ref = torch.empty((3, 3, 3, 32), dtype=torch.float64)
    ref.conv_wgrad(x.double(), dy_ref, dw_ref, 1)
    # HIP: forward statistics, then the fused backward
    xd, yd, dad, gd, bd = dev(x), dev(y), dev(da), dev(gamma), dev(beta)
    a, st = torch.empty((B, H, W, 32), device="cuda"), torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(yd, gd, bd, a, st)
    ws = torch.zeros(hip.ln_workspace_bytes((B, H, W, 32)), dtype=torch.uint8, device="cuda")      # (zeroed: compared as a whole below)
    means = torch.full((B, 2), float("nan"), device="cuda")
    hip.ln_elu_bwd_sums(yd, dad, gd, bd, st, means, ws)
    dw = torch.full((3, 3, 3, 32), float("nan"), device="cuda")
    hip.conv_c3_wgrad_ln(xd, yd, dad, gd, bd, st, means, dw)
    close(dw, dw_ref, rtol=5e-6, what="fused conv1_1 filter gradient vs fp64")
    # the unfused path on the same operands
    ws2 = torch.zeros_like(ws)
    dy = torch.empty((B, H, W, 32), device="cuda")
    hip.ln_elu_bwd(yd, dad, gd, bd, st, dy, None, None, None, ws=ws2)
    dw2 = torch.empty((3, 3, 3, 32), device="cuda")
    old = hip.conv_precision
    hip.conv_precision = 0
    try:
        hip.conv_wgrad(xd, dy, dw2, 1)
    finally:
        hip.conv_precision = old
    close(dw, dw2, rtol=2e-6, what="fused vs unfused conv1_1 filter gradient")
    assert torch.equal(ws, ws2), "the partial sums of the reductions-only entry point differ from sgg_layernorm_hwc_elu_bwd's"
    # twice the same launch: bit-identical (fixed summation order)
    dw3 = torch.empty_like(dw)
    hip.conv_c3_wgrad_ln(xd, yd, dad, gd, bd, st, means, dw3)
    assert torch.equal(dw, dw3)
    # the per-sample means are what the apply pass derives: dy recomputed on the host from them equals the apply pass's dy
    m = means.cpu().double()
    s = st.cpu().double()
    xh = (y.double() - s[:, 0].view(B, 1, 1, 1)) * s[:, 1].view(B, 1, 1, 1)
    n = xh * gamma.double() + beta.double()
    dn = da.double() * torch.where(n > 0, torch.ones_like(n), torch.exp(n))
    dy_host = s[:, 1].view(B, 1, 1, 1) * (dn * gamma.double() - m[:, 0].view(B, 1, 1, 1) - xh * m[:, 1].view(B, 1, 1, 1))
    close(dy, dy_host, rtol=5e-6, what="dy from the published means vs the apply pass")
