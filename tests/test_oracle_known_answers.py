"""CPU: pins the oracle (oracle/sgg_oracle.py).  The reference cannot run in this pipeline and holds no tests or
golden vectors, so the oracle is pinned by (1) hand-derived known answers for every TF-1.x rule in
SURVEY.md Appendix A, (2) an independent plain-loop NumPy restatement (oracle/np_loops.py), (3) fp64 finite
differences of the full critic loss incl. the gradient penalty (the double-backward ground truth)."""
import math

import numpy as np
import pytest
import torch

from oracle import np_loops as NL
from oracle import sgg_oracle as O

T64 = torch.float64


def test_same_padding_rule_A1():
    # (out, before, after): 5x5 stride 2 on even sizes -> (1,2); on 221 / 111 -> (2,2); 3x3 stride 1 -> (1,1)
    assert O.same_pads(224, 5, 2) == (112, 1, 2)
    assert O.same_pads(56, 5, 2) == (28, 1, 2)
    assert O.same_pads(221, 5, 2) == (111, 2, 2)
    assert O.same_pads(111, 5, 2) == (56, 2, 2)
    assert O.same_pads(224, 3, 1) == (224, 1, 1)
    assert O.feature_side(224) == 14 and O.feature_side(64) == 4 and O.feature_side(448) == 28 and O.feature_side(221) == 14


def test_conv_same_asymmetric_known_answer():
    # 4x4 ones, 5x5 ones kernel, stride 2: pad (1,2) -> windows cover rows {-1..3} -> 4 valid, {1..5} -> 3 valid
    x = torch.ones((1, 4, 4, 1), dtype=T64)
    w = torch.ones((5, 5, 1, 1), dtype=T64)
    y = O.conv2d_same(x, w, torch.tensor([0.5], dtype=T64), 2)
    assert y.shape == (1, 2, 2, 1)
    assert torch.equal(y[0, :, :, 0], torch.tensor([[16.5, 12.5], [12.5, 9.5]], dtype=T64))
    # cross-correlation (no kernel flip): a one-hot kernel tap picks the pixel at that offset
    x = torch.arange(9, dtype=T64).reshape(1, 3, 3, 1)
    w = torch.zeros((3, 3, 1, 1), dtype=T64)
    w[0, 2] = 1.0      # tap (kh=0, kw=2) -> reads x[h-1, w+1]
    y = O.conv2d_same(x, w, torch.zeros(1, dtype=T64), 1)
    assert float(y[0, 1, 1, 0]) == 2.0 and float(y[0, 2, 0, 0]) == 4.0 and float(y[0, 0, 0, 0]) == 0.0


def test_layer_norm_elu_known_answer_A2():
    x = torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]], dtype=T64)     # [1,1,2,2]: mean 2.5, biased var 1.25
    y = O.layer_norm_tf(x, torch.tensor([2.0, 0.5], dtype=T64), torch.tensor([1.0, -1.0], dtype=T64))
    r = 1.0 / math.sqrt(1.25)
    exp = torch.tensor([[[[1 - 3 * r, -1 - 0.25 * r], [1 + r, -1 + 0.75 * r]]]], dtype=T64)   # xhat = (-1.5r, -.5r, .5r, 1.5r)
    assert torch.allclose(y, exp, atol=1e-12)
    assert abs(float(O.elu(torch.tensor(-1.0, dtype=T64))) - (math.exp(-1) - 1)) < 1e-15
    assert float(O.elu(torch.tensor(2.0, dtype=T64))) == 2.0


def test_lnlstm_known_answer_A5():
    n, B = 512, 2
    p = {"layer_norm_basic_lstm_cell/kernel": torch.zeros((1024 + n, 4 * n), dtype=T64)}
    betas = {"input": 50.0, "transform": 0.5, "forget": -100.0, "output": 0.0, "state": 0.3}
    for sc, bv in betas.items():
        p["layer_norm_basic_lstm_cell/%s/gamma" % sc] = torch.ones(n, dtype=T64)
        p["layer_norm_basic_lstm_cell/%s/beta" % sc] = torch.full((n,), bv, dtype=T64)
    x, c, h = torch.randn(B, 1024, dtype=T64), torch.randn(B, n, dtype=T64), torch.randn(B, n, dtype=T64)
    new_h, new_c = O.lnlstm_cell(p, x, c, h)
    # zero kernel -> gates = LN(0) = beta: i ~ 1, j -> tanh(.5), f -> sigmoid(-99) = 0: c' = LN(const) = beta_state
    assert torch.allclose(new_c, torch.full((B, n), 0.3, dtype=T64), atol=1e-6)
    assert torch.allclose(new_h, torch.full((B, n), math.tanh(0.3) * 0.5, dtype=T64), atol=1e-6)


def test_tf_adam_epsilon_placement_A8():
    for g0 in (1.0, 1e-8):
        th, g = {"w": torch.tensor([0.0], dtype=T64)}, {"w": torch.tensor([g0], dtype=T64)}
        m, v = {"w": torch.zeros(1, dtype=T64)}, {"w": torch.zeros(1, dtype=T64)}
        O.tf_adam_step(th, g, m, v, 1)
        lr_t = 1e-4 * math.sqrt(0.1) / 0.5
        exp = -lr_t * 0.5 * g0 / (math.sqrt(0.1) * g0 + 1e-8)
        assert abs(float(th["w"]) - exp) < 1e-18
    # eps OUTSIDE the bias correction: for g = 1e-8 TF moves 2.40e-5, torch.optim.Adam's form would move 5e-5
    assert abs(exp + 2.4025e-5) < 1e-8
    n1, *_ = NL.tf_adam(np.zeros(1), np.array([1e-8]), np.zeros(1), np.zeros(1), 1)
    assert abs(n1[0] - exp) < 1e-18


def test_argmax_first_index_A9():
    x = torch.tensor([[1.0, 5.0, 5.0, 2.0], [0.0, 0.0, 0.0, 0.0], [-1.0, -3.0, -1.0, -1.0]])
    assert O.argmax_tokens(x).tolist() == [1, 0, 0]
    assert NL.argmax_first(x.numpy()).tolist() == [1, 0, 0]


def test_numpy_loop_restatement_agrees():
    g = torch.Generator().manual_seed(0)
    for (H, W, Ci, Co, k, s) in [(5, 6, 3, 4, 3, 1), (6, 6, 2, 3, 5, 2), (7, 5, 2, 2, 5, 2)]:
        x, w, b = (torch.randn(s_, generator=g, dtype=T64) for s_ in ((2, H, W, Ci), (k, k, Ci, Co), (Co,)))
        assert np.allclose(O.conv2d_same(x, w, b, s).numpy(), NL.conv2d_same(x.numpy(), w.numpy(), b.numpy(), s), atol=1e-12)
    x, ga, be = torch.randn((3, 4, 4, 8), generator=g, dtype=T64), torch.randn(8, generator=g, dtype=T64), torch.randn(8, generator=g, dtype=T64)
    assert np.allclose(O.layer_norm_tf(x, ga, be).numpy(), NL.layer_norm(x.numpy(), ga.numpy(), be.numpy()), atol=1e-10)
    assert np.allclose(O.elu(x).numpy(), NL.elu(x.numpy()), atol=1e-14)
    # LSTM cell and attention with the real parameter layout
    n, ind, L, B = 512, 812, 4, 2
    p = {"layer_norm_basic_lstm_cell/kernel": torch.randn((ind + n, 4 * n), generator=g, dtype=T64) * 0.05}
    ln = []
    for sc in O.LSTM_LN_SCOPES:
        ga, be = 1 + 0.2 * torch.randn(n, generator=g, dtype=T64), 0.2 * torch.randn(n, generator=g, dtype=T64)
        p["layer_norm_basic_lstm_cell/%s/gamma" % sc], p["layer_norm_basic_lstm_cell/%s/beta" % sc] = ga, be
        ln.append((ga.numpy(), be.numpy()))
    x, c, h = (torch.randn(s_, generator=g, dtype=T64) for s_ in ((B, ind), (B, n), (B, n)))
    nh, nc = O.lnlstm_cell(p, x, c, h)
    nh2, nc2 = NL.lnlstm_cell(x.numpy(), c.numpy(), h.numpy(), p["layer_norm_basic_lstm_cell/kernel"].numpy(), ln)
    assert np.allclose(nh.numpy(), nh2, atol=1e-10) and np.allclose(nc.numpy(), nc2, atol=1e-10)
    ctx = torch.randn((B, L, 512), generator=g, dtype=T64)
    pa = {"attention_perceptron/kernel": torch.randn((L * 512 + 512, L), generator=g, dtype=T64) * 0.05,
          "attention_perceptron/bias": torch.randn(L, generator=g, dtype=T64)}
    z, al = O.attention(pa, ctx.reshape(B, -1), ctx, c)
    z2, al2 = NL.attention(ctx.numpy(), c.numpy(), pa["attention_perceptron/kernel"].numpy(), pa["attention_perceptron/bias"].numpy())
    assert np.allclose(z.numpy(), z2, atol=1e-10) and np.allclose(al.numpy(), al2, atol=1e-12)


def test_gradient_penalty_one_sided_A7():
    from oracle.kernels_ref import RefKernels
    g = torch.zeros((2, 3, 4), dtype=T64)
    g[0, 0, 0] = 3.0            # slope 3 -> penalty (3-1)^2 = 4
    g[1, 1, 1] = 0.5            # slope 0.5 -> one-sided: 0
    sl, pen = torch.empty(2, dtype=T64), torch.empty(2, dtype=T64)
    RefKernels().gp_fwd(g, sl, pen)
    assert abs(float(sl[0]) - math.sqrt(9 + 1e-10)) < 1e-12 and abs(float(pen[0]) - (math.sqrt(9 + 1e-10) - 1)) < 1e-12
    assert float(pen[1]) == 0.0
    assert abs(float((pen ** 2).mean()) - 2.0) < 1e-9


def test_full_critic_loss_gradient_by_finite_differences():
    """d disc_cost / d theta incl. the gradient penalty: autograd double-backward vs central differences (fp64)."""
    B, S, V = 2, 32, 7
    gp, dp = O.init_params("G", V, S, dtype=T64, perturb=0.1), O.init_params("D", V, S, dtype=T64, perturb=0.1)
    dp["W"] = dp["W"] * 25.0
    images, _, onehot = O.synth_batch(B, S, V, dtype=T64)
    noise, alpha = O.synth_noise(B, 0, T64), O.synth_alpha(B, 0, T64)
    names = ["W", "layer_norm_basic_lstm_cell/kernel", "layer_norm_basic_lstm_cell/forget/gamma", "attention_perceptron/kernel",
             "conv2d_13/kernel", "LayerNorm_9/beta", "decoder/kernel"]
    for v in dp.values():
        v.requires_grad_(True)
    cost, aux = O.d_loss(gp, dp, images, onehot, noise, alpha, 10.0)
    assert float(aux["gp"]) > 1e-3
    grads = torch.autograd.grad(cost, [dp[n] for n in names])
    for v in dp.values():
        v.requires_grad_(False)
    gen = torch.Generator().manual_seed(9)
    for n, g in zip(names, grads):
        d = torch.randn(dp[n].shape, generator=gen, dtype=T64)
        d = d / d.norm()
        eps = 1e-5
        orig = dp[n].clone()
        dp[n] = orig + eps * d
        cp, _ = O.d_loss(gp, dp, images, onehot, noise, alpha, 10.0)
        dp[n] = orig - eps * d
        cm, _ = O.d_loss(gp, dp, images, onehot, noise, alpha, 10.0)
        dp[n] = orig
        fd = float(cp - cm) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)) + 1e-7, "%s: finite difference %.8e vs autograd %.8e" % (n, fd, an)
