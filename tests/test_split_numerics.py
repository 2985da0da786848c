"""CPU: emulation of the split 16-bit convolution arithmetic (csrc/split16.h, DESIGN.md "Split 16-bit convolution modes")
against fp64, next to plain f32 - the numerics claim behind the default conv mode, as a test.

The MFMA contracts 16-bit pieces exactly (a product of two 11-bit or 8-bit significands fits f32) and accumulates in
f32, so what distinguishes a split mode from native f32 is (i) the rounding of each operand into its pieces and (ii) the
dropped cross terms.  The emulation makes exactly those two approximations and then contracts in fp64:
    mode 2 (f16x3)  : x*2^e = hi + lo, hi = rne_f16(x*2^e), lo = rne_f16(x*2^e - hi); products hi*hi + hi*lo + lo*hi
    mode 6 (bf16x6) : x = x0 + x1 + x2 (bf16, rne); the six products of total order <= 2
    mode 3 (bf16x3) : x = x0 + x1; three products
e is the per-tensor power of two with max|x|*2^e <= 2^14 (csrc/sgg_common.h: scale_exp_from_amax).
Reference points: conv in fp64 (truth) and conv in f32 (what TensorFlow's fp32 kernels deliver).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import sgg_oracle as O


def scale_exp(t):
    amax = float(t.abs().max())
    if amax <= 0:
        return 0
    _, k = math.frexp(amax)
    return max(-100, min(100, 14 - k))


def pieces_f16(t):
    e = scale_exp(t)
    s = t * (2.0 ** e)
    hi = s.half().float()
    lo = (s - hi).half().float()
    return [hi.double(), lo.double()], 2.0 ** (-e)


def pieces_bf16(t, n):
    out, r = [], t.clone()
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p.double())
        r = r - p
    return out, 1.0


def conv64(x, w, s):
    return O.conv2d_same(x, w, torch.zeros(w.shape[3], dtype=x.dtype), s)


def split_conv(x, w, s, mode):
    if mode == 2:
        (xp, sx), (wp, sw) = pieces_f16(x), pieces_f16(w)
    else:
        (xp, sx), (wp, sw) = pieces_bf16(x, 3 if mode == 6 else 2), pieces_bf16(w, 3 if mode == 6 else 2)
    order = 1 if mode in (2, 3) else 2
    y = 0
    for i, a in enumerate(xp):
        for j, b in enumerate(wp):
            if i + j <= order:
                y = y + conv64(a, b, s)
    return y * (sx * sw)


def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("k,s", [(3, 1), (5, 2)])
def test_single_layer_split_error_is_at_f32_rounding_level(k, s):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 24, 24, 64), generator=g)
    w = torch.randn((k, k, 64, 64), generator=g) * math.sqrt(2.0 / (k * k * 64))
    ref = conv64(x.double(), w.double(), s)
    e32 = rel(conv64(x, w, s), ref)
    e2, e6, e3 = (rel(split_conv(x, w, s, m), ref) for m in (2, 6, 3))
    print("k%d s%d: f32 %.2e  f16x3 %.2e  bf16x6 %.2e  bf16x3 %.2e" % (k, s, e32, e2, e6, e3))
    assert e2 <= 2 * e32 + 1e-7 and e6 <= 2 * e32 + 1e-7        # f32-equivalent modes
    assert e3 <= 1e-4                                            # two bf16 pieces: the path's stated tolerance
    assert e2 < 3e-7 and e6 < 3e-7


def test_heavy_tailed_operand_f16x3():
    """Per-tensor scaling: an element 2^-d below the tensor maximum keeps min(23, 39 - d) bits; error stays at f32 level
    relative to the tensor as long as the bulk sits within 2^17 of the maximum."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 16, 16, 64), generator=g)
    x.view(-1)[1234] = 1e4                                       # bulk is 2^13 below the maximum
    w = torch.randn((3, 3, 64, 64), generator=g) * 0.05
    ref = conv64(x.double(), w.double(), 1)
    e32, e2 = rel(conv64(x, w, 1), ref), rel(split_conv(x, w, 1, 2), ref)
    assert e2 <= 2 * e32 + 1e-7
    x2 = torch.randn((2, 16, 16, 64), generator=g)
    x2[1] *= 1e-6                                                # a whole sample 2^-20 below: ~17 bits on its own scale
    ref2 = conv64(x2.double(), w.double(), 1)
    y2 = split_conv(x2, w, 1, 2)
    assert rel(y2, ref2) <= 2 * rel(conv64(x2, w, 1), ref2) + 1e-7
    own = float((y2[1] - ref2[1]).abs().max() / ref2[1].abs().max())
    print("sample scaled by 1e-6: error on its own scale %.2e" % own)
    assert own <= 1e-4


def test_encoder_stack_split_matches_f32_accuracy():
    """Four conv + LayerNorm + ELU layers (the encoder's pattern, generator_with_attention.py:29-47): error of the stack
    output against fp64 with every convolution in f32, f16x3 and bf16x3 arithmetic."""
    g = torch.Generator().manual_seed(3)
    chans = [(32, 32, 3, 1), (32, 32, 5, 2), (32, 64, 3, 1), (64, 64, 3, 1)]
    ws = [torch.randn((k, k, ci, co), generator=g) * math.sqrt(2.0 / (k * k * ci)) for ci, co, k, s in chans]
    x0 = torch.randn((2, 32, 32, 32), generator=g)

    def stack(conv, dtype):
        a = x0.to(dtype)
        for (ci, co, k, s), w in zip(chans, ws):
            y = conv(a, w, s).to(dtype) + 0.05
            a = O.elu(O.layer_norm_tf(y, torch.ones(co, dtype=dtype), torch.zeros(co, dtype=dtype)))
        return a

    ref = stack(lambda a, w, s: conv64(a.double(), w.double(), s), torch.float64)
    e32 = rel(stack(lambda a, w, s: conv64(a, w, s), torch.float32), ref)
    e2 = rel(stack(lambda a, w, s: split_conv(a, w, s, 2), torch.float32), ref)
    e3 = rel(stack(lambda a, w, s: split_conv(a, w, s, 3), torch.float32), ref)
    print("4-layer stack vs fp64: f32 %.2e  f16x3 %.2e  bf16x3 %.2e" % (e32, e2, e3))
    assert e2 <= 2 * e32 + 2e-7
    assert e3 <= 1e-4
