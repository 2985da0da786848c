"""CPU ORACLE (test infrastructure, NOT product code).

A plain PyTorch-CPU restatement of the reference's WGAN-GP training hot path, written from the
reference's Python sources and the TensorFlow-1.x op semantics they invoke.

PARITY UNPINNED: the reference (Python 2 + TensorFlow 1.x `tf.contrib`) cannot be imported or run in
this pipeline (no TensorFlow, no Python 2; `train.py` is a SyntaxError under Python 3) and it holds no
tests, fixtures or golden vectors.  The restatement is therefore pinned only by hand-derived
known-answer tests (tests/test_oracle_known_answers.py), an independent NumPy-loop restatement of the
small ops (oracle/np_loops.py) and fp64 finite-difference checks.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this module.
The product path (scene-graph-gan_amd/, architectures/, train.py) never does.

All randomness (noise, alpha, initial weights) is an explicit input, so results are reproducible and
comparable with the HIP path bit-for-bit in their inputs.

Reference citations (relative to /root/reference):
  encoder            architectures/generator_with_attention.py:21-68 (discriminator_with_attention.py:21-68)
  context / state    generator_with_attention.py:74-77
  attentionMechanism generator_with_attention.py:13-18
  G step loop        generator_with_attention.py:79-91
  D step loop        discriminator_with_attention.py:81-93
  loss / optimiser   train.py:239-266
  argmax probe       train.py:270-271
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# Architecture table.  One row per `tf.layers.conv2d` in definition order (the TF default layer
# names are numbered in this order, dead layers included: SURVEY.md Appendix A.10).
#   (index, cin, cout, k, stride, has_layernorm, live)
# generator_with_attention.py:29-68
# ---------------------------------------------------------------------------------------------
CONV_SPECS = (
    (0, 3, 32, 3, 1, True, True),      # conv1_1   :29
    (1, 32, 32, 3, 1, True, True),     # conv1_2   :31
    (2, 32, 32, 5, 2, True, True),     # conv1_3   :35
    (3, 32, 64, 3, 1, True, True),     # conv2_1   :39
    (4, 64, 64, 3, 1, True, True),     # conv2_2   :41
    (5, 64, 128, 3, 1, True, True),    # conv2_3   :44
    (6, 128, 128, 3, 1, True, True),   # conv2_4   :46
    (7, 128, 128, 5, 2, True, True),   # conv2_5   :50
    (8, 128, 256, 3, 1, True, True),   # conv3_1   :54
    (9, 256, 256, 3, 1, True, True),   # conv3_2   :56
    (10, 256, 512, 3, 1, True, False),  # conv3_3  :59  DEAD (output never reaches `downsampled`)
    (11, 512, 512, 3, 1, True, False),  # conv3_4  :61  DEAD
    (12, 256, 512, 5, 2, True, True),  # conv3_5   :65  (input = layernorm3_2, i.e. layer 9's output)
    (13, 512, 512, 5, 2, False, True),  # downsampled :68 (no LN / ELU)
)
LIVE_CONVS = tuple(s for s in CONV_SPECS if s[6])
NUM_UNITS = 512          # LayerNormBasicLSTMCell(512), generator_with_attention.py:79
FEAT_C = 512
T_STEPS = 3              # range(3), generator_with_attention.py:85
LN_EPS = 1e-12           # tf.contrib.layers.layer_norm variance_epsilon
FORGET_BIAS = 1.0        # LayerNormBasicLSTMCell default
LSTM_LN_SCOPES = ("input", "transform", "forget", "output", "state")


def conv_name(i: int) -> str:
    return "conv2d" if i == 0 else "conv2d_%d" % i


def ln_name(i: int) -> str:
    return "LayerNorm" if i == 0 else "LayerNorm_%d" % i


def same_pads(in_size: int, k: int, s: int) -> Tuple[int, int, int]:
    """TF 'SAME' padding (Appendix A.1): returns (out, pad_before, pad_after)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def feature_side(S: int) -> int:
    h = S
    for (_, _, _, k, s, _, live) in CONV_SPECS:
        if live:
            h = same_pads(h, k, s)[0]
    return h


# ---------------------------------------------------------------------------------------------
# Primitive ops with TF-1.x semantics
# ---------------------------------------------------------------------------------------------
def conv2d_same(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, stride: int) -> torch.Tensor:
    """tf.layers.conv2d(padding="same"): x NHWC, w HWIO, cross-correlation, bias added after (A.1)."""
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pads(x.shape[1], kh, stride)
    _, pl, pr = same_pads(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def layer_norm_tf(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """tf.contrib.layers.layer_norm defaults (A.2): statistics over every axis but the batch axis,
    gamma/beta over the last axis, biased two-pass variance, eps=1e-12, batch_normalization form."""
    dims = tuple(range(1, x.dim()))
    mean = x.mean(dim=dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
    inv = torch.rsqrt(var + LN_EPS) * gamma
    return x * inv + (beta - mean * inv)


def elu(x: torch.Tensor) -> torch.Tensor:
    return torch.where(x > 0, x, torch.expm1(torch.clamp(x, max=0.0)))


def encoder(p: Dict[str, torch.Tensor], images: torch.Tensor, return_all: bool = False):
    """12 live conv layers (generator_with_attention.py:29-68).  Dead layers 10/11 are never computed."""
    acts = {}
    x = images
    for (i, cin, cout, k, s, has_ln, live) in CONV_SPECS:
        if not live:
            continue
        # layer 12 (conv3_5) reads layernorm3_2 = output of layer 9 — which is simply the running `x`,
        # because the dead layers 10/11 are skipped (generator_with_attention.py:65).
        y = conv2d_same(x, p[conv_name(i) + "/kernel"], p[conv_name(i) + "/bias"], s)
        if has_ln:
            x = elu(layer_norm_tf(y, p[ln_name(i) + "/gamma"], p[ln_name(i) + "/beta"]))
        else:
            x = y
        if return_all:
            acts[i] = (y, x)
    return (x, acts) if return_all else x


def attention(p, ctx_flat: torch.Tensor, ctx: torch.Tensor, c: torch.Tensor):
    """attentionMechanism (generator_with_attention.py:13-18): dense over [flattened map, c] -> softmax
    over the L locations -> weighted sum of the feature rows.  `cell_state[0]` is c (A.5)."""
    e = torch.cat([ctx_flat, c], dim=1) @ p["attention_perceptron/kernel"] + p["attention_perceptron/bias"]
    alpha = torch.softmax(e, dim=1)
    z = (ctx * alpha.unsqueeze(2)).sum(dim=1)
    return z, alpha


def lnlstm_cell(p, x: torch.Tensor, c: torch.Tensor, h: torch.Tensor):
    """tf.contrib.rnn.LayerNormBasicLSTMCell(512) step (A.5). Gate order i, j, f, o; no bias; LN per gate;
    new_c = LN(c*sigmoid(f+1) + sigmoid(i)*tanh(j)); new_h = tanh(new_c)*sigmoid(o)."""
    pre = "layer_norm_basic_lstm_cell/"
    concat = torch.cat([x, h], dim=1) @ p[pre + "kernel"]
    i, j, f, o = torch.chunk(concat, 4, dim=1)
    i = layer_norm_tf(i, p[pre + "input/gamma"], p[pre + "input/beta"])
    j = layer_norm_tf(j, p[pre + "transform/gamma"], p[pre + "transform/beta"])
    f = layer_norm_tf(f, p[pre + "forget/gamma"], p[pre + "forget/beta"])
    o = layer_norm_tf(o, p[pre + "output/gamma"], p[pre + "output/beta"])
    g = torch.tanh(j)
    new_c = c * torch.sigmoid(f + FORGET_BIAS) + torch.sigmoid(i) * g
    new_c = layer_norm_tf(new_c, p[pre + "state/gamma"], p[pre + "state/beta"])
    new_h = torch.tanh(new_c) * torch.sigmoid(o)
    return new_h, new_c


def context_views(downsampled: torch.Tensor):
    """generator_with_attention.py:74-77."""
    B, H, W, C = downsampled.shape
    ctx_flat = downsampled.reshape(B, H * W * C)
    ctx = downsampled.reshape(B, H * W, C)
    m = downsampled.mean(dim=(1, 2))
    return ctx_flat, ctx, m


def generator_head(p, downsampled: torch.Tensor, noise: torch.Tensor, return_aux: bool = False):
    """generator_with_attention.py:74-91 given the feature map."""
    ctx_flat, ctx, m = context_views(downsampled)
    c, h = m, m
    logits, alphas = [], []
    for _ in range(T_STEPS):
        z, alpha = attention(p, ctx_flat, ctx, c)
        x = torch.cat([z, noise], dim=1)
        h, c = lnlstm_cell(p, x, c, h)
        logits.append(h @ p["decoder/kernel"] + p["decoder/bias"])
        alphas.append(alpha)
    out = torch.stack(logits, dim=1)
    return (out, alphas) if return_aux else out


def generator_forward(p, images: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """Generator.build_generator (generator_with_attention.py:20-91) -> logits [B,3,V]."""
    return generator_head(p, encoder(p, images), noise)


def discriminator_head(p, downsampled: torch.Tensor, triples: torch.Tensor) -> torch.Tensor:
    """discriminator_with_attention.py:73-93 given the feature map. p["W"] is the embedding matrix."""
    ctx_flat, ctx, m = context_views(downsampled)
    c, h = m, m
    outs = []
    for t in range(T_STEPS):
        emb = triples[:, t, :] @ p["W"]
        z, _ = attention(p, ctx_flat, ctx, c)
        x = torch.cat([z, emb], dim=1)
        h, c = lnlstm_cell(p, x, c, h)
        outs.append(h @ p["decoder/kernel"] + p["decoder/bias"])
    return torch.stack(outs, dim=1)


def discriminator_forward(p, triples: torch.Tensor, images: torch.Tensor) -> torch.Tensor:
    """Discriminator.build_discriminator (discriminator_with_attention.py:20-93) -> [B,3,1]."""
    return discriminator_head(p, encoder(p, images), triples)


# ---------------------------------------------------------------------------------------------
# Losses (train.py:239-250; tfgan semantics: Appendix A.6 / A.7)
# ---------------------------------------------------------------------------------------------
GP_EPS = 1e-10


def gradient_penalty(dp, feat_d: torch.Tensor, real: torch.Tensor, fake: torch.Tensor, alpha: torch.Tensor,
                     create_graph: bool = True):
    """tfgan wasserstein_gradient_penalty(one_sided=True, target=1, epsilon=1e-10)."""
    xhat = real + alpha * (fake - real)
    if not xhat.requires_grad:
        xhat = xhat.detach().requires_grad_(True)
    d_hat = discriminator_head(dp, feat_d, xhat)
    (g,) = torch.autograd.grad(d_hat.sum(), xhat, create_graph=create_graph)
    slopes = torch.sqrt((g ** 2).sum(dim=(1, 2)) + GP_EPS)
    pen = torch.clamp(slopes - 1.0, min=0.0)
    return (pen ** 2).mean(), slopes, g


def d_loss(gp, dp, images, onehot, noise, alpha, lam: float):
    """disc_cost of train.py:245-253.  The D optimiser only touches Discriminator* variables
    (train.py:263,266), so the fake triples are a constant here."""
    with torch.no_grad():
        fake = generator_forward(gp, images, noise)
    feat_d = encoder(dp, images)            # same images for fake / real / interpolated
    d_fake = discriminator_head(dp, feat_d, fake)
    d_real = discriminator_head(dp, feat_d, onehot)
    gpen, slopes, _ = gradient_penalty(dp, feat_d, onehot, fake, alpha)
    wdist = d_fake.mean() - d_real.mean()
    cost = wdist + lam * gpen
    return cost, {"fake": fake, "d_fake": d_fake, "d_real": d_real, "gp": gpen, "slopes": slopes, "wdist": wdist}


def g_loss(gp, dp, images, noise):
    """gen_cost of train.py:247,252: -mean(D(G(x)))."""
    fake = generator_forward(gp, images, noise)
    d_fake = discriminator_forward(dp, fake, images)
    return -d_fake.mean(), {"fake": fake, "d_fake": d_fake}


def argmax_tokens(logits: torch.Tensor) -> torch.Tensor:
    """tf.argmax(x, -1) (train.py:270): int64, first index on ties (A.9)."""
    V = logits.shape[-1]
    mx = logits.max(dim=-1, keepdim=True).values
    idx = torch.arange(V, dtype=torch.int64).expand_as(logits)
    return torch.where(logits == mx, idx, torch.full_like(idx, V)).min(dim=-1).values


def top2_margin(logits: torch.Tensor) -> float:
    top = logits.topk(2, dim=-1).values
    return float((top[..., 0] - top[..., 1]).min())


# ---------------------------------------------------------------------------------------------
# tf.train.AdamOptimizer (train.py:258-259; A.8)
# ---------------------------------------------------------------------------------------------
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 1e-4, 0.5, 0.9, 1e-8


def tf_adam_lr_t(t: int, lr=ADAM_LR, b1=ADAM_B1, b2=ADAM_B2) -> float:
    return lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)


def tf_adam_step(params, grads, m, v, t: int, lr=ADAM_LR, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
    """In-place TF-1.x Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps).
    Variables whose gradient is None (dead branch) are skipped and get no slots.  `t` is the 1-based step."""
    lr_t = tf_adam_lr_t(t, lr, b1, b2)
    lr_t = torch.tensor(lr_t, dtype=torch.float64)
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            dt = params[k].dtype
            m[k].mul_(b1).add_(g, alpha=1.0 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            params[k].sub_(lr_t.to(dt) * m[k] / (v[k].sqrt() + eps))


# ---------------------------------------------------------------------------------------------
# Parameter construction (TF variable names relative to the network scope; A.10)
# ---------------------------------------------------------------------------------------------
def _trunc_normal(gen, shape, std, dtype):
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t.to(dtype)


def _glorot_uniform(gen, shape, dtype):
    lim = math.sqrt(6.0 / (shape[0] + shape[1]))
    return ((torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * lim).to(dtype)


def param_shapes(kind: str, V: int, S: int, E: int = 300) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered name -> shape map for one network ('G' or 'D'), dead-branch variables included."""
    assert kind in ("G", "D")
    hs = feature_side(S)
    L = hs * hs
    sh = OrderedDict()
    for (i, cin, cout, k, s, has_ln, live) in CONV_SPECS:
        sh[conv_name(i) + "/kernel"] = (k, k, cin, cout)
        sh[conv_name(i) + "/bias"] = (cout,)
        if has_ln:
            sh[ln_name(i) + "/gamma"] = (cout,)
            sh[ln_name(i) + "/beta"] = (cout,)
    sh["attention_perceptron/kernel"] = (L * FEAT_C + NUM_UNITS, L)
    sh["attention_perceptron/bias"] = (L,)
    in_dim = FEAT_C + (NUM_UNITS if kind == "G" else E)
    sh["layer_norm_basic_lstm_cell/kernel"] = (in_dim + NUM_UNITS, 4 * NUM_UNITS)
    for sc in LSTM_LN_SCOPES:
        sh["layer_norm_basic_lstm_cell/%s/gamma" % sc] = (NUM_UNITS,)
        sh["layer_norm_basic_lstm_cell/%s/beta" % sc] = (NUM_UNITS,)
    out_dim = V if kind == "G" else 1
    sh["decoder/kernel"] = (NUM_UNITS, out_dim)
    sh["decoder/bias"] = (out_dim,)
    if kind == "D":
        sh["W"] = (V, E)     # train.py:68-72: owned by the trainer, trained by D's optimiser
    return sh


def is_dead(name: str) -> bool:
    return name.split("/")[0] in ("conv2d_10", "conv2d_11", "LayerNorm_10", "LayerNorm_11")


def init_params(kind: str, V: int, S: int, E: int = 300, seed: int = 3, dtype=torch.float32,
                perturb: float = 0.0) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic initial weights: conv kernels truncated-normal he (sigma=sqrt(2/fan_in)),
    conv bias 0.05 (generator_with_attention.py:21-22), LN gamma 1 / beta 0, dense & LSTM kernels
    Glorot-uniform, dense bias 0, embedding U(-0.1,0.1) (map_files_to_triples.py:24).
    `perturb` > 0 adds N(0, perturb) to gamma/beta/bias so tests exercise non-trivial values."""
    gen = torch.Generator().manual_seed(seed + (0 if kind == "G" else 1000))
    p = OrderedDict()
    for name, shape in param_shapes(kind, V, S, E).items():
        leaf = name.split("/")[-1]
        if name.startswith("conv2d") and leaf == "kernel":
            fan_in = shape[0] * shape[1] * shape[2]
            t = _trunc_normal(gen, shape, math.sqrt(2.0 / fan_in), dtype)
        elif name.startswith("conv2d") and leaf == "bias":
            t = torch.full(shape, 0.05, dtype=dtype)
        elif leaf == "gamma":
            t = torch.ones(shape, dtype=dtype)
        elif leaf == "beta":
            t = torch.zeros(shape, dtype=dtype)
        elif leaf == "kernel":
            t = _glorot_uniform(gen, shape, dtype)
        elif leaf == "bias":
            t = torch.zeros(shape, dtype=dtype)
        elif name == "W":
            t = ((torch.rand(shape, generator=gen, dtype=torch.float32) * 0.2) - 0.1).to(dtype)
        else:
            raise KeyError(name)
        if perturb > 0 and leaf in ("gamma", "beta", "bias"):
            t = t + perturb * torch.randn(shape, generator=gen, dtype=torch.float32).to(dtype)
        p[name] = t
    return p


# ---------------------------------------------------------------------------------------------
# Synthetic inputs
# ---------------------------------------------------------------------------------------------
def synth_batch(B: int, S: int, V: int, seed_img: int = 0, seed_lab: int = 1, dtype=torch.float32):
    gi = torch.Generator().manual_seed(seed_img)
    gl = torch.Generator().manual_seed(seed_lab)
    images = torch.randn((B, S, S, 3), generator=gi, dtype=torch.float32).to(dtype)
    labels = torch.randint(0, V, (B, T_STEPS), generator=gl, dtype=torch.int64)
    onehot = F.one_hot(labels, V).to(dtype)
    return images, labels, onehot


def synth_noise(B: int, step: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(2 + step)
    return torch.randn((B, NUM_UNITS), generator=g, dtype=torch.float32).to(dtype)


def synth_alpha(B: int, step: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + step)
    return torch.rand((B, 1, 1), generator=g, dtype=torch.float32).to(dtype)


# ---------------------------------------------------------------------------------------------
# One full G+D step on the oracle (train.py:362-368 with CRITIC_ITERS critic updates)
# ---------------------------------------------------------------------------------------------
def _grads(cost, params) -> Dict[str, torch.Tensor]:
    names = [k for k in params if not is_dead(k)]
    gs = torch.autograd.grad(cost, [params[k] for k in names], allow_unused=True)
    return {k: g for k, g in zip(names, gs)}


def new_adam_state(params):
    m = {k: torch.zeros_like(v) for k, v in params.items() if not is_dead(k)}
    v = {k: torch.zeros_like(v) for k, v in params.items() if not is_dead(k)}
    return m, v


def d_step(gp, dp, d_adam, t, images, onehot, noise, alpha, lam=10.0):
    for v in dp.values():
        v.requires_grad_(True)
    cost, aux = d_loss(gp, dp, images, onehot, noise, alpha, lam)
    grads = _grads(cost, dp)
    for v in dp.values():
        v.requires_grad_(False)
    tf_adam_step(dp, grads, d_adam[0], d_adam[1], t)
    return cost.detach(), aux, grads


def g_step(gp, dp, g_adam, t, images, noise):
    for v in gp.values():
        v.requires_grad_(True)
    cost, aux = g_loss(gp, dp, images, noise)
    grads = _grads(cost, gp)
    for v in gp.values():
        v.requires_grad_(False)
    tf_adam_step(gp, grads, g_adam[0], g_adam[1], t)
    return cost.detach(), aux, grads
