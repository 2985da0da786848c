"""fp64 convolution references for the full-size GPU parity tests (test infrastructure).

Restates tf.layers.conv2d(padding="same") (generator_with_attention.py:29-68) and its two gradients
(Conv2DBackpropInput / Conv2DBackpropFilter, as optimizer.minimize produces them, train.py:265-266) as one fp64 matrix
product per kernel tap over shifted, strided slices of the zero-padded input.  Same arithmetic as
oracle/sgg_oracle.py::conv2d_same and its autograd gradients (pinned against them in tests/test_conv_ref64.py), but fast
enough on CPU for the B = 64 layer shapes of BASELINE.json configs[1] (wgrad contraction lengths up to 3.2 M)."""
import torch


def same_pads(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def _padded(x, k, s):
    B, H, W, C = x.shape
    Ho, pt, pb = same_pads(H, k, s)
    Wo, pl, pr = same_pads(W, k, s)
    xp = torch.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    return xp, Ho, Wo, pt, pl


def _tap(xp, kh, kw, Ho, Wo, s):
    return xp[:, kh:kh + (Ho - 1) * s + 1:s, kw:kw + (Wo - 1) * s + 1:s, :]


def conv_fwd64(x, w, bias, s):
    """x [B,H,W,Ci], w [k,k,Ci,Co] (HWIO), bias [Co] or None -> y [B,Ho,Wo,Co]; all fp64."""
    k, _, Ci, Co = w.shape
    xp, Ho, Wo, _, _ = _padded(x, k, s)
    B = x.shape[0]
    y = torch.zeros((B * Ho * Wo, Co), dtype=x.dtype)
    for kh in range(k):
        for kw in range(k):
            y += _tap(xp, kh, kw, Ho, Wo, s).reshape(-1, Ci) @ w[kh, kw]
    if bias is not None:
        y += bias
    return y.view(B, Ho, Wo, Co)


def conv_wgrad64(x, dy, k, s):
    """dw [k,k,Ci,Co] = sum over (b, y, x) of x_tap^T dy."""
    B, H, W, Ci = x.shape
    Co = dy.shape[3]
    xp, Ho, Wo, _, _ = _padded(x, k, s)
    assert tuple(dy.shape[:3]) == (B, Ho, Wo)
    dy2 = dy.reshape(-1, Co)
    dw = torch.empty((k, k, Ci, Co), dtype=x.dtype)
    for kh in range(k):
        for kw in range(k):
            dw[kh, kw] = _tap(xp, kh, kw, Ho, Wo, s).reshape(-1, Ci).t() @ dy2
    return dw


def conv_dgrad64(dy, w, in_hw, s):
    """dx [B,H,W,Ci] from dy [B,Ho,Wo,Co] and the HWIO kernel."""
    k, _, Ci, Co = w.shape
    B, Ho, Wo, _ = dy.shape
    H, W = in_hw
    xp, Ho2, Wo2, pt, pl = _padded(torch.zeros((B, H, W, Ci), dtype=dy.dtype), k, s)
    assert (Ho, Wo) == (Ho2, Wo2)
    dy2 = dy.reshape(-1, Co)
    for kh in range(k):
        for kw in range(k):
            _tap(xp, kh, kw, Ho, Wo, s).add_((dy2 @ w[kh, kw].t()).view(B, Ho, Wo, Ci))
    return xp[:, pt:pt + H, pl:pl + W, :].contiguous()
