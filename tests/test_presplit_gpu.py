"""-m gpu: pre-split ("S16") activations - the LayerNorm kernels write a = ELU(LN(y)) and dy of the LayerNorm backward as two fp16
pieces per value inside the f32 tensor's bytes (csrc/split16.h; include/sgg_hip.h out_format / operand_format), and the resident
convolution kernels stage such operands without splitting them again.

  * producers: the planes are bit for bit what today's staging split (sgg_common.h: f16_split2, round to nearest even twice) makes
    of the f32 output, with the scale taken from the published BOUND of max|x|; the bound holds and is tight to a few binades;
  * consumers (every resident kernel family: four-wave 3x3 in its 32 / 64 / 128-column tilings, producer / consumer 3x3, band-resident
    5x5 stride 2, conv1_3 through the space-to-depth view; filter gradients on 8x8 blocks, in the stride-2 parity classes and in row
    bands): with a pre-split operand the outputs are BIT-EQUAL to the f32 operand's (same pieces, same products, same order).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32) * scale


def s16_exp(amax):
    """scale_exp_from_amax of sgg_common.h."""
    if not amax > 0.0:
        return 0
    _, k = math.frexp(amax)
    return max(-100, min(100, 14 - k))


def to_s16(x, amax):
    """Host emulation of split8<2, true> into the S16 layout: x [..., C] f32 (C % 32 == 0) -> f32-typed tensor of the same shape whose
    bytes are, per aligned 32-channel group, 32 leading fp16 pieces then 32 residual pieces of x * 2^e."""
    C = x.shape[-1]
    xs = x.float() * (2.0 ** s16_exp(amax))
    hi = xs.half()
    lo = (xs - hi.float()).half()
    out = torch.stack([hi.reshape(-1, C // 32, 32), lo.reshape(-1, C // 32, 32)], dim=2)      # [px, group, 2, 32]
    return out.contiguous().view(torch.int16).reshape(-1).view(torch.float32).reshape(x.shape)


def s16_pieces(t):
    """S16 tensor (f32-typed) -> (hi, lo) as int16 bit patterns [px, C / 32, 32]."""
    C = t.shape[-1]
    h = t.contiguous().view(torch.int16).reshape(-1, C // 32, 2, 32)
    return h[:, :, 0, :], h[:, :, 1, :]


def from_s16(t, amax):
    C = t.shape[-1]
    h = t.contiguous().view(torch.float16).reshape(-1, C // 32, 2, 32).float()
    return ((h[:, :, 0, :] + h[:, :, 1, :]) * (2.0 ** -s16_exp(amax))).reshape(t.shape)


@pytest.fixture
def mode2(hip):
    old = hip.conv_precision
    hip.conv_precision = 2
    yield hip
    hip.conv_precision = old


LN_SHAPES = [((3, 7, 7, 32), None), ((2, 9, 9, 64), None),      # ragged tails: the last lanes of a wave have no neighbour group
             ((2, 16, 16, 32), None), ((3, 8, 8, 128), None), ((2, 40, 40, 64), None), ((1, 8, 8, 512), None), ((2, 16, 16, 32), (1, 1, 15, 15)),
             ((3, 24, 24, 64), (3, 3, 21, 21))]


@pytest.mark.parametrize("shape,region", LN_SHAPES)
def test_layernorm_outputs_presplit_are_todays_split_bit_for_bit(mode2, shape, region):
    hip = mode2
    B, H, W, C = shape
    y, da = rnd(shape, 1).cuda() * 1.7 + 0.3, rnd(shape, 2).cuda()
    gamma, beta = (1.0 + rnd((C,), 3, 0.2)).cuda(), rnd((C,), 4, 0.2).cuda()
    # ---- forward: a = ELU(LN(y)) ----
    a32, a16 = torch.empty_like(y), torch.full_like(y, float("nan"))
    st32, st16 = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
    w32, w16 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    hip.ln_elu_fwd(y, gamma, beta, a32, st32, w32, None, region=region)
    hip.ln_elu_fwd(y, gamma, beta, a16, st16, w16, None, region=region, out_s16=True)
    assert torch.allclose(st16, st32, rtol=1e-6, atol=0)
    amax, bound = float(w32), float(w16)
    assert amax <= bound <= 16.0 * amax, (amax, bound)          # an upper bound, within four binades
    exp_hi, exp_lo = s16_pieces(to_s16(a32.cpu(), bound))
    got_hi, got_lo = s16_pieces(a16.cpu())
    assert torch.equal(got_hi, exp_hi) and torch.equal(got_lo, exp_lo), "forward planes differ from f16_split2 of the f32 output"
    rec = from_s16(a16.cpu(), bound)
    assert float((rec - a32.cpu()).abs().max()) <= 2.0 ** -21 * bound
    # ---- backward: dy ----
    dy32, dy16 = torch.empty_like(y), torch.full_like(y, float("nan"))
    ws = torch.empty(hip.ln_workspace_bytes(shape), dtype=torch.uint8, device="cuda")
    v32, v16 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    hip.ln_elu_bwd(y, da, gamma, beta, st32, dy32, None, None, None, v32, region=region, ws=ws)
    hip.ln_elu_bwd(y, da, gamma, beta, st32, dy16, None, None, None, v16, region=region, ws=ws, out_s16=True)
    amax, bound = float(v32), float(v16)
    assert amax <= bound <= 64.0 * amax, (amax, bound)
    exp_hi, exp_lo = s16_pieces(to_s16(dy32.cpu(), bound))
    got_hi, got_lo = s16_pieces(dy16.cpu())
    assert torch.equal(got_hi, exp_hi) and torch.equal(got_lo, exp_lo), "backward planes differ from f16_split2 of the f32 output"


def _weights(hip, Ci, Co, k, lay_f, lay_b, seed):
    w = (rnd((k, k, Ci, Co), seed, 1.0 / math.sqrt(k * k * Ci))).cuda()
    wf = torch.empty((k, k, Co, Ci), device="cuda")
    hip.hwio_to_hwoi(w, wf)
    am = torch.zeros(1, device="cuda")
    hip.absmax(w, am)
    ws_f = torch.empty((3, w.numel() * (4 if 3 in (lay_f, lay_b) else 1)), dtype=torch.int16, device="cuda")
    ws_b = torch.empty_like(ws_f)
    if lay_f == 3:
        w3, w3f = torch.empty((3, 3, 4 * Ci, Co), device="cuda"), torch.empty((3, 3, Co, 4 * Ci), device="cuda")
        hip.s2d_weights(w, w3)
        hip.hwio_to_hwoi(w3, w3f)
        hip.split_weights(w3f, ws_f, am, 3)
        hip.split_weights(w3, ws_b, am, 3)
    else:
        hip.split_weights(wf, ws_f, am, lay_f)
        hip.split_weights(w, ws_b, am, lay_b)
    return w, wf, am, ws_f, ws_b


@pytest.mark.parametrize("shape", [(2, 14, 14, 512), (3, 5, 7, 32), (1, 1, 1, 96)])
def test_presplit16_conversion_is_the_split(mode2, shape):
    """sgg_presplit16: an f32 tensor that no LayerNorm kernel produces (the head's gradient w.r.t. `downsampled`) -> the same
    pre-split format, bit for bit f16_split2 under the scale of the amax word (a maximum or a bound); also in place."""
    hip = mode2
    x = rnd(shape, 41, 3.0)
    x[0, 0, 0, 0] = 17.0
    bits = lambda t: t.contiguous().view(torch.int32)
    am = torch.zeros(1, device="cuda")
    xd = x.cuda()
    hip.absmax(xd, am)
    assert float(am) == float(x.abs().max())
    out = torch.full(shape, float("nan"), device="cuda")
    hip.presplit16(xd, out, am)
    assert torch.equal(bits(out.cpu()), bits(to_s16(x, float(am))))
    hip.presplit16(xd, xd, am)                                    # in place
    assert torch.equal(bits(xd), bits(out))
    bound = torch.tensor([2.7 * float(am)], device="cuda")        # a bound instead of the maximum: another scale, same rule
    hip.presplit16(x.cuda(), out, bound)
    assert torch.equal(bits(out.cpu()), bits(to_s16(x, 2.7 * float(am))))


FWD_CASES = [
    # B, H, W, Cin, Cout, k, stride      (the resident kernel families)
    (3, 16, 24, 32, 32, 3, 1),       # four-wave 3x3, 32-column tiling (two-wave workgroups)
    (2, 16, 16, 32, 64, 3, 1),       # 64-column tiling
    (2, 16, 16, 64, 64, 3, 1),
    (2, 16, 16, 64, 128, 3, 1),      # producer / consumer kernel forward (layout 4), 64-column dgrad
    (2, 8, 16, 128, 128, 3, 1),      # producer / consumer both ways
    (1, 8, 8, 256, 256, 3, 1),
    (2, 32, 32, 32, 32, 5, 2),       # conv1_3 through the space-to-depth view (layout 3)
    (2, 16, 16, 128, 128, 5, 2),     # band-resident 5x5 stride 2 (layout 2)
    (2, 16, 16, 256, 512, 5, 2),
    (4, 8, 8, 512, 512, 5, 2),       # few work items: channel chunks split over two workgroups
    # 256+ output columns and more than 128 (band, 256-column) items: eight-wave workgroups, one patch per 256 columns
    (10, 112, 112, 256, 256, 5, 2),  # 56 x 56 grid: 140 bands, wide in both directions
    (5, 112, 112, 128, 512, 5, 2),   # 70 bands x two 256-column tiles forward; 128 columns in the dgrad direction
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_convolutions_with_presplit_operands_are_bit_equal(mode2, case):
    hip = mode2
    B, H, W, Ci, Co, k, s = case
    from sgg_amd.lib import same_pads
    Ho, Wo = same_pads(H, k, s)[0], same_pads(W, k, s)[0]
    lay_f, lay_b = hip.conv_wsplit_layout(k, s, H, W, Ci, Co), hip.conv_wsplit_layout(k, s, H, W, Co, Ci)
    assert lay_f in (1, 2, 3, 4) and lay_b in (1, 2, 3, 4), (lay_f, lay_b)
    w, wf, am_w, ws_f, ws_b = _weights(hip, Ci, Co, k, lay_f, lay_b, 7)
    x, dy, b = rnd((B, H, W, Ci), 5), rnd((B, Ho, Wo, Co), 6), rnd((Co,), 8, 0.1).cuda()
    # the words hold BOUNDS (what the LayerNorm kernels publish), not the maxima
    bx, bdy = 1.9 * float(x.abs().max()), 3.3 * float(dy.abs().max())
    am_x, am_dy = torch.tensor([bx], device="cuda"), torch.tensor([bdy], device="cuda")
    x32, dy32, x16, dy16 = x.cuda(), dy.cuda(), to_s16(x, bx).cuda(), to_s16(dy, bdy).cuda()
    # forward
    y_a, y_b = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda"), torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    hip.conv_fwd(x32, w, wf, b, y_a, s, ws_f, am_x, am_w, None, lay_f)
    hip.conv_fwd(x16, w, wf, b, y_b, s, ws_f, am_x, am_w, None, lay_f, x_s16=True)
    assert torch.isfinite(y_a).all() and torch.equal(y_a, y_b), "forward, layout %d: max |d| = %.3e" % (lay_f, float((y_a - y_b).abs().max()))
    # dgrad
    dx_a, dx_b = torch.full((B, H, W, Ci), float("nan"), device="cuda"), torch.full((B, H, W, Ci), float("nan"), device="cuda")
    hip.conv_dgrad(dy32, w, dx_a, s, ws_b, am_dy, am_w, lay_b)
    hip.conv_dgrad(dy16, w, dx_b, s, ws_b, am_dy, am_w, lay_b, dy_s16=True)
    assert torch.isfinite(dx_a).all() and torch.equal(dx_a, dx_b), "dgrad, layout %d: max |d| = %.3e" % (lay_b, float((dx_a - dx_b).abs().max()))
    # filter gradient: either operand, and both
    assert hip.wgrad_resident(B, Ho, Wo, Ci, Co, k, s)
    dw = [torch.full((k, k, Ci, Co), float("nan"), device="cuda") for _ in range(4)]
    hip.conv_wgrad(x32, dy32, dw[0], s, am_x, am_dy)
    hip.conv_wgrad(x16, dy32, dw[1], s, am_x, am_dy, x_s16=True)
    hip.conv_wgrad(x32, dy16, dw[2], s, am_x, am_dy, dy_s16=True)
    hip.conv_wgrad(x16, dy16, dw[3], s, am_x, am_dy, x_s16=True, dy_s16=True)
    assert torch.isfinite(dw[0]).all()
    for i in (1, 2):        # one pre-split operand: the register-staged kernel, same blocks per workgroup -> the same sums
        assert torch.equal(dw[0], dw[i]), "wgrad variant %d: max |d| = %.3e" % (i, float((dw[0] - dw[i]).abs().max()))
    # both pre-split: the LDS-DMA kernel where it takes the shape (same products, another partition of the blocks), else as above
    if H % (8 * s) == 0 and W % (8 * s) == 0 and Ci % 64 == 0 and Co % 64 == 0:
        assert float((dw[0] - dw[3]).abs().max()) <= 5e-6 * float(dw[0].abs().max())
    else:
        assert torch.equal(dw[0], dw[3])


WGRAD_BAND_CASES = [(2, 56, 56, 64, 128, 5, 2), (3, 28, 28, 128, 64, 5, 2), (1, 14, 14, 64, 64, 5, 2), (3, 20, 20, 64, 64, 3, 1), (1, 9, 9, 64, 128, 3, 1)]


@pytest.mark.parametrize("case", WGRAD_BAND_CASES)
def test_row_band_filter_gradient_with_presplit_operands_is_bit_equal(mode2, case):
    hip = mode2
    B, H, W, Ci, Co, k, s = case
    from sgg_amd.lib import same_pads
    Ho, Wo = same_pads(H, k, s)[0], same_pads(W, k, s)[0]
    if not (same_pads(H, k, s)[1] == 1 and hip.wgrad_resident(B, Ho, Wo, Ci, Co, k, s)):
        pytest.skip("shape not on the halo-resident filter-gradient kernel")
    x, dy = rnd((B, H, W, Ci), 5), rnd((B, Ho, Wo, Co), 6)
    bx, bdy = 1.3 * float(x.abs().max()), 5.0 * float(dy.abs().max())
    am_x, am_dy = torch.tensor([bx], device="cuda"), torch.tensor([bdy], device="cuda")
    dw = [torch.full((k, k, Ci, Co), float("nan"), device="cuda") for _ in range(4)]
    x16, dy16 = to_s16(x, bx).cuda(), to_s16(dy, bdy).cuda()
    hip.conv_wgrad(x.cuda(), dy.cuda(), dw[0], s, am_x, am_dy)
    hip.conv_wgrad(x16, dy.cuda(), dw[1], s, am_x, am_dy, x_s16=True)                    # register-staged row-band kernel: bit-equal
    hip.conv_wgrad(x.cuda(), dy16, dw[2], s, am_x, am_dy, dy_s16=True)
    hip.conv_wgrad(x16, dy16, dw[3], s, am_x, am_dy, x_s16=True, dy_s16=True)            # LDS-DMA row-band kernel: another summation order
    assert torch.isfinite(dw[0]).all() and torch.equal(dw[0], dw[1]) and torch.equal(dw[0], dw[2])
    assert torch.isfinite(dw[3]).all()
    assert float((dw[0] - dw[3]).abs().max()) <= 5e-6 * float(dw[0].abs().max()), "max |d| = %.3e" % float((dw[0] - dw[3]).abs().max())


def test_presplit_operands_are_rejected_where_no_kernel_takes_them(hip):
    from sgg_amd.lib import SggError
    old = hip.conv_precision
    try:
        x, w = torch.zeros((1, 8, 8, 32), device="cuda"), torch.zeros((3, 3, 32, 32), device="cuda")
        wf, b, y = torch.zeros((3, 3, 32, 32), device="cuda"), torch.zeros(32, device="cuda"), torch.zeros((1, 8, 8, 32), device="cuda")
        hip.conv_precision = 0
        with pytest.raises(SggError):
            hip.conv_fwd(x, w, wf, b, y, 1, x_s16=True)                       # native f32 MFMA has no pieces
        hip.conv_precision = 2
        am = torch.ones(1, device="cuda")
        with pytest.raises(SggError):
            hip.conv_fwd(x, w, wf, b, y, 1, None, am, am, None, 0, x_s16=True)   # the gather kernel (layout 0) splits f32 itself
        with pytest.raises(SggError):
            hip.ln_elu_fwd(y, b, b, y.clone(), torch.zeros((1, 2), device="cuda"), None, None, out_s16=True)     # needs the amax word
    finally:
        hip.conv_precision = old


DMA_CASES = [
    # B, H, W, Cin, Cout, k, stride: shapes the LDS-DMA filter-gradient kernel takes (8-divisible dy grid, channels % 64 == 0)
    (2, 16, 16, 64, 128, 3, 1),      # 64 x 128 channel tile, 8 blocks: fewer blocks than workgroup slots
    (3, 24, 16, 128, 128, 3, 1),     # two input-channel tiles, border blocks on every side
    (2, 8, 8, 256, 256, 3, 1),       # one block per image: every halo row / column is padding
    (5, 40, 24, 64, 64, 3, 1),       # 64 x 64 tile (two pixel halves per workgroup), 75 blocks
    (1, 8, 16, 128, 64, 3, 1),
    (3, 32, 32, 128, 128, 5, 2),     # the four stride-2 parity classes (conv2_5)
    (2, 16, 48, 64, 128, 5, 2),
    (3, 56, 56, 64, 128, 5, 2),      # row bands: 28 x 28 grid in 4-row bands (conv3_5)
    (2, 28, 28, 128, 64, 5, 2),      # 14 x 14 grid in 8-row bands, the second band ragged (`downsampled`)
    (2, 20, 20, 64, 64, 3, 1),       # 20 x 20: bands of 5 rows
    (1, 9, 9, 64, 128, 3, 1),        # one band per image
]


@pytest.mark.parametrize("case", DMA_CASES)
def test_dma_filter_gradient_matches_fp64_and_the_register_staged_kernel(mode2, ref, case):
    """conv_wgrad_dma_kernel (both operands pre-split, staged by LDS-DMA: out-of-range lanes must arrive as zeros) against the fp64
    reference and against the halo-resident kernel on the same pieces (same products; only the partition of the blocks differs)."""
    hip = mode2
    B, H, W, Ci, Co, k, s = case
    from sgg_amd.lib import same_pads
    Ho, Wo = same_pads(H, k, s)[0], same_pads(W, k, s)[0]
    x, dy = rnd((B, H, W, Ci), 5), rnd((B, Ho, Wo, Co), 6)
    # spikes on the image border: a halo lane that is not zeroed would multiply them
    x[:, 0, :, :] *= 3.0
    x[:, :, -1, :] *= 3.0
    bx, bdy = 1.9 * float(x.abs().max()), 3.3 * float(dy.abs().max())
    am_x, am_dy = torch.tensor([bx], device="cuda"), torch.tensor([bdy], device="cuda")
    dw_ref = torch.empty((k, k, Ci, Co), dtype=torch.float64)
    # (the reference sees the values the pieces represent: the split is exact to 2^-22 of the bound)
    ref.conv_wgrad(from_s16(to_s16(x, bx), bx).double(), from_s16(to_s16(dy, bdy), bdy).double(), dw_ref, s)
    x16, dy16 = to_s16(x, bx).cuda(), to_s16(dy, bdy).cuda()
    dw_dma = torch.full((k, k, Ci, Co), float("nan"), device="cuda")
    hip.conv_wgrad(x16, dy16, dw_dma, s, am_x, am_dy, x_s16=True, dy_s16=True)          # both pre-split: the DMA kernel
    dw_reg = torch.full((k, k, Ci, Co), float("nan"), device="cuda")
    hip.conv_wgrad(x16, dy.cuda(), dw_reg, s, am_x, am_dy, x_s16=True)                  # one f32 operand: the register-staged kernel
    scale = float(dw_ref.abs().max())
    assert torch.isfinite(dw_dma).all()
    e_dma, e_reg = float((dw_dma.cpu().double() - dw_ref).abs().max()), float((dw_reg.cpu().double() - dw_ref).abs().max())
    assert e_dma <= 2e-5 * scale, "DMA kernel vs fp64: %.3e of %.3e" % (e_dma, scale)
    assert e_reg <= 2e-5 * scale
    assert float((dw_dma - dw_reg).abs().max()) <= 5e-6 * scale
    # twice the same launch: bit-identical (fixed summation order)
    dw2 = torch.empty_like(dw_dma)
    hip.conv_wgrad(x16, dy16, dw2, s, am_x, am_dy, x_s16=True, dy_s16=True)
    assert torch.equal(dw2, dw_dma)


@pytest.mark.parametrize("sigma", [0.0, 50.0, 400.0], ids=["gaussian", "pixel_at_50_sigma", "pixel_at_400_sigma"])
def test_presplit_dy_bound_under_outlier_activations(mode2, ref, sigma):
    """The scale of a pre-split dy (LayerNorm backward) comes from the bound 1.0001 * P * (2 + Q), P = max rstd * max|dx_hat|,
    Q = max|x_hat| (csrc/layernorm.hip): with heavy-tailed pre-activations (one pixel tens of sigmas out, as real images produce) Q
    makes the bound k binades larger than max|dy|, and every binade takes one bit from the SMALL elements of dy (an element 2^-d below
    the maximum keeps min(23, 39 - d - k) bits).  Measured here on a conv2_2-sized layer (4 x 112 x 112 x 64): k, and what it does to
    the filter gradient that consumes this dy by LDS-DMA - against fp64 and against the same kernel on the EXACT maximum (k = 0).
    The degradation must stay inside the f32 path's own bound (2e-5 of the gradient's maximum; tests/test_fullsize_conv_gpu.py)."""
    import json
    import os
    hip = mode2
    B, H, C, Co = 4, 112, 64, 64
    y, da = rnd((B, H, H, C), 11), rnd((B, H, H, C), 12)
    if sigma:
        y[1, 40, 57, :] = sigma                   # one pixel, all channels, `sigma` standard deviations out
        y[3, 0, 0, 5] = -0.6 * sigma
    x = rnd((B, H, H, C), 13)                     # the activation the filter gradient contracts dy with (conv 3x3, 64 -> 64)
    gamma, beta = (1.0 + rnd((C,), 3, 0.2)).cuda(), rnd((C,), 4, 0.2).cuda()
    yd, dad = y.cuda(), da.cuda()
    a = torch.empty_like(yd)
    st = torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(yd, gamma, beta, a, st, torch.zeros(1, device="cuda"), None)
    ws = torch.empty(hip.ln_workspace_bytes((B, H, H, C)), dtype=torch.uint8, device="cuda")
    dy32, dy16 = torch.empty_like(yd), torch.empty_like(yd)
    v32, v16 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    hip.ln_elu_bwd(yd, dad, gamma, beta, st, dy32, None, None, None, v32, ws=ws)
    hip.ln_elu_bwd(yd, dad, gamma, beta, st, dy16, None, None, None, v16, ws=ws, out_s16=True)
    amax, bound = float(v32), float(v16)
    k = math.log2(bound / amax)
    assert amax <= bound, (amax, bound)
    # the filter gradient on (x pre-split under its exact maximum, dy pre-split under the bound) against fp64 of the f32 tensors
    amx = x.abs().max().reshape(1).cuda()
    x16 = torch.empty((B, H, H, C), device="cuda")
    hip.presplit16(x.cuda(), x16, amx)
    dw_ref = torch.empty((3, 3, C, Co), dtype=torch.float64)
    ref.conv_wgrad(x.double(), dy32.cpu().double(), dw_ref, 1)
    scale = float(dw_ref.abs().max())
    dw_b = torch.empty((3, 3, C, Co), device="cuda")
    hip.conv_wgrad(x16, dy16, dw_b, 1, amx, v16, x_s16=True, dy_s16=True)
    dyx = torch.empty_like(yd)
    hip.presplit16(dy32, dyx, v32)               # the same tensor split under its exact maximum: k = 0
    dw_x = torch.empty((3, 3, C, Co), device="cuda")
    hip.conv_wgrad(x16, dyx, dw_x, 1, amx, v32, x_s16=True, dy_s16=True)
    e_b = float((dw_b.cpu().double() - dw_ref).abs().max()) / scale
    e_x = float((dw_x.cpu().double() - dw_ref).abs().max()) / scale
    # the small elements of dy on their own scale: the error of the split itself where |dy| <= 2^-12 max|dy|
    small = dy32.abs() <= amax * 2.0 ** -12
    rec = from_s16(dy16.cpu(), bound).cuda()
    own = float(((rec - dy32).abs()[small] / (dy32.abs()[small] + 1e-30)).max()) if bool(small.any()) else 0.0
    print("outlier %5.0f sigma: max|x_hat| %.1f  bound / max|dy| = 2^%.2f   wgrad err bound %.3e  exact-max %.3e   worst relative split error of "
          "elements <= 2^-12 max: %.3e" % (sigma, float(((yd - st[:, 0].view(B, 1, 1, 1)) * st[:, 1].view(B, 1, 1, 1)).abs().max()), k, e_b, e_x, own))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        p = os.path.join(d, "presplit_dy_bound_outliers.json")
        allr = json.load(open(p)) if os.path.exists(p) else {}
        allr["%g_sigma" % sigma] = {"k_binades": k, "wgrad_err_under_bound": e_b, "wgrad_err_under_exact_max": e_x, "small_element_rel_split_err": own}
        json.dump(allr, open(p, "w"), indent=1)
    assert k <= 11.0, "the bound leaves fewer than 5 of the 16 spare binades (k = %.2f)" % k
    assert e_b <= 2e-5 and e_b <= 2.0 * e_x + 3e-7, (e_b, e_x)


PC64_CASES = [
    # B, H, W, C (contraction channels), N = 64 output columns: the four-block form of the producer / consumer kernel
    (2, 16, 16, 64),
    (1, 8, 24, 128),        # three blocks: the fourth block of the only tile is dead
    (3, 24, 40, 64),        # 45 blocks: a ragged last tile
    (5, 8, 8, 256),         # 5 blocks in 2 tiles; eight chunks per tile
    (32, 56, 56, 64),       # 1568 blocks = 392 tiles: 49 per XCD on 32 workgroups - several tiles per workgroup (conv2_2's dgrad shape)
]


@pytest.mark.parametrize("case", PC64_CASES)
def test_producer_consumer_four_block_tiles(mode2, ref, case):
    """conv_halo3_pc_kernel<true, false, true, 4> (round 5): 4 blocks x 64 columns per workgroup, reached through w_split_layout 4 with
    Cout % 64 == 0 and a PRE-SPLIT source (sgg_conv_wsplit_layout_presplit).  Forward (bias, LayerNorm tile statistics) and dgrad
    against fp64 and against the four-wave kernel on the same pieces (same products; the order inside a 32-channel chunk differs);
    an f32 source with that layout must be refused."""
    from sgg_amd.lib import SggError
    hip = mode2
    B, H, W, C = case
    N = 64
    assert hip.conv_wsplit_layout(3, 1, H, W, C, N) == 1 and hip.conv_wsplit_layout_presplit(3, 1, H, W, C, N) == 4
    assert hip.conv_wsplit_layout_presplit(3, 1, H, W, C, 128) == 4 and hip.conv_wsplit_layout_presplit(3, 1, H, W, 32, N) == 1
    x, w, b = rnd((B, H, W, C), 21), rnd((3, 3, C, N), 22, 1.0 / math.sqrt(9 * C)), rnd((N,), 23, 0.1)
    # spikes on the image border: a halo lane that is not zeroed would multiply them
    x[:, 0, :, :] *= 3.0
    x[:, :, -1, :] *= 3.0
    bx = 1.7 * float(x.abs().max())
    am_x = torch.tensor([bx], device="cuda")
    xq = from_s16(to_s16(x, bx), bx) if B * H * W * C <= (1 << 22) else None      # (the values the pieces represent: exact to 2^-22 of the bound)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    x16 = torch.empty_like(xd)
    hip.presplit16(xd, x16, am_x)
    am_w = torch.zeros(1, device="cuda")
    hip.absmax(wd, am_w)
    # ---- forward: y = conv(x, w) + b, N = 64 output columns ----
    wf = torch.empty((3, 3, N, C), device="cuda")
    hip.hwio_to_hwoi(wd, wf)
    ws4, ws1 = (torch.empty((2, w.numel()), dtype=torch.int16, device="cuda") for _ in range(2))
    hip.split_weights(wf, ws4, am_w, layout=4)
    hip.split_weights(wf, ws1, am_w, layout=1)
    nts = hip.conv_tile_stats_count((B, H, W, N), C, 3, 1, 4)
    assert nts == H * W // 64
    ts = torch.full((B, nts, 4), float("nan"), device="cuda")
    y4 = torch.full((B, H, W, N), float("nan"), device="cuda")
    hip.conv_fwd(x16, wd, wf, bd, y4, 1, ws4, am_x, am_w, ts, 4, x_s16=True)
    y1 = torch.empty_like(y4)
    hip.conv_fwd(x16, wd, wf, bd, y1, 1, ws1, am_x, am_w, None, 1, x_s16=True)
    assert torch.isfinite(y4).all()
    scale = float(y1.abs().max())
    assert float((y4 - y1).abs().max()) <= 5e-6 * scale, "four-block producer / consumer forward vs the four-wave kernel"
    if xq is not None:
        y_ref = torch.empty((B, H, W, N), dtype=torch.float64)
        ref.conv_fwd(xq.double(), w.double(), None, b.double(), y_ref, 1)
        assert float((y4.cpu().double() - y_ref).abs().max()) <= 2e-5 * float(y_ref.abs().max())
    g4, b4 = (1.0 + rnd((N,), 15, 0.2)).cuda(), rnd((N,), 16, 0.2).cuda()
    a_t, a_p = torch.empty_like(y4), torch.empty_like(y4)
    st_t, st_p = torch.empty((B, 2), device="cuda"), torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(y4, g4, b4, a_t, st_t, tile_stats=ts)
    hip.ln_elu_fwd(y4, g4, b4, a_p, st_p)
    # (mean: absolute against the sample's scale 1 / rstd; rstd: relative)
    assert float((st_t[:, 0] - st_p[:, 0]).abs().max()) <= 1e-6 / float(st_p[:, 1].min()) and \
        float(((st_t[:, 1] - st_p[:, 1]) / st_p[:, 1]).abs().max()) <= 1e-6, "tile statistics of the four-block epilogue vs the statistics pass"
    # ---- dgrad: dx [.., 64] = conv-transpose of dy [.., C] with w [3,3,64,C] ----
    w2 = rnd((3, 3, N, C), 24, 1.0 / math.sqrt(9 * C))
    w2d = w2.cuda()
    am_w2 = torch.zeros(1, device="cuda")
    hip.absmax(w2d, am_w2)
    wb4, wb1 = (torch.empty((2, w2.numel()), dtype=torch.int16, device="cuda") for _ in range(2))
    hip.split_weights(w2d, wb4, am_w2, layout=4)
    hip.split_weights(w2d, wb1, am_w2, layout=1)
    assert hip.conv_wsplit_layout_presplit(3, 1, H, W, C, N) == 4      # (dgrad is asked with the channels swapped: contraction C, output N)
    dx4 = torch.full((B, H, W, N), float("nan"), device="cuda")
    hip.conv_dgrad(x16, w2d, dx4, 1, wb4, am_x, am_w2, 4, dy_s16=True)
    dx1 = torch.empty_like(dx4)
    hip.conv_dgrad(x16, w2d, dx1, 1, wb1, am_x, am_w2, 1, dy_s16=True)
    assert torch.isfinite(dx4).all()
    assert float((dx4 - dx1).abs().max()) <= 5e-6 * float(dx1.abs().max()), "four-block producer / consumer dgrad vs the four-wave kernel"
    if xq is not None:
        dx_ref = torch.empty((B, H, W, N), dtype=torch.float64)
        ref.conv_dgrad(xq.double(), w2.double(), dx_ref, 1)
        assert float((dx4.cpu().double() - dx_ref).abs().max()) <= 2e-5 * float(dx_ref.abs().max())
    # twice the same launch: bit-identical
    dx5 = torch.empty_like(dx4)
    hip.conv_dgrad(x16, w2d, dx5, 1, wb4, am_x, am_w2, 4, dy_s16=True)
    assert torch.equal(dx5, dx4)
    # an f32 source has no four-block variant: refused, loudly
    with pytest.raises(SggError):
        hip.conv_dgrad(xd, w2d, dx5, 1, wb4, am_x, am_w2, 4)
