"""Architecture table and parameter arenas of the two networks, with the reference's TF variable names.

Reference: architectures/generator_with_attention.py:21-91, architectures/discriminator_with_attention.py:21-93,
train.py:65-72 (embedding variable `Discriminator/W`); naming rules SURVEY.md Appendix A.10.

Every trainable tensor of a network is a view into ONE flat fp32 arena (live parameters first, the dead
conv3_3/conv3_4 branch at the tail), and gradients / Adam slots use arenas of the same layout.  That makes the
optimiser one fused elementwise kernel over a contiguous range and the data-parallel all-reduce a handful of
large buckets (MI355X: few, large collectives over xGMI).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

# (index, cin, cout, k, stride, has_layernorm, live) in definition order  (generator_with_attention.py:29-68)
CONV_SPECS = (
    (0, 3, 32, 3, 1, True, True),
    (1, 32, 32, 3, 1, True, True),
    (2, 32, 32, 5, 2, True, True),
    (3, 32, 64, 3, 1, True, True),
    (4, 64, 64, 3, 1, True, True),
    (5, 64, 128, 3, 1, True, True),
    (6, 128, 128, 3, 1, True, True),
    (7, 128, 128, 5, 2, True, True),
    (8, 128, 256, 3, 1, True, True),
    (9, 256, 256, 3, 1, True, True),
    (10, 256, 512, 3, 1, True, False),   # conv3_3: dead (generator_with_attention.py:59)
    (11, 512, 512, 3, 1, True, False),   # conv3_4: dead (:61)
    (12, 256, 512, 5, 2, True, True),    # conv3_5 reads layernorm3_2 (:65)
    (13, 512, 512, 5, 2, False, True),   # downsampled (:68)
)
NUM_UNITS = 512
FEAT_C = 512
T_STEPS = 3
EMBED_DIM = 300
LSTM_LN_SCOPES = ("input", "transform", "forget", "output", "state")
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 1e-4, 0.5, 0.9, 1e-8     # train.py:258-259 (+ TF default epsilon)


def conv_name(i):
    return "conv2d" if i == 0 else "conv2d_%d" % i


def ln_name(i):
    return "LayerNorm" if i == 0 else "LayerNorm_%d" % i


def same_pads(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def feature_side(S):
    h = S
    for (_, _, _, k, s, _, live) in CONV_SPECS:
        if live:
            h = same_pads(h, k, s)[0]
    return h


def is_dead(name):
    return name.split("/")[0] in ("conv2d_10", "conv2d_11", "LayerNorm_10", "LayerNorm_11")


def param_shapes(kind, V, S, E=EMBED_DIM):
    """Ordered TF-name -> shape map ('G' or 'D'), dead-branch variables included."""
    assert kind in ("G", "D")
    L = feature_side(S) ** 2
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
        sh["W"] = (V, E)
    return sh


def tf_variable_name(kind, name):
    """Full TF-1.x variable name of the trained variable set (Appendix A.10)."""
    scope = "Generator" if kind == "G" else "Discriminator"
    if name == "W":
        return "Discriminator/W"
    return "%s/%s/%s" % (scope, scope, name)


class ParamArena:
    """Flat fp32 storage for one network: .flat (live range first), .views[name] -> shaped view."""

    def __init__(self, kind, V, S, E=EMBED_DIM, device="cpu", dtype=torch.float32):
        self.kind, self.V, self.S, self.E = kind, V, S, E
        self.shapes = param_shapes(kind, V, S, E)
        names_live = [n for n in self.shapes if not is_dead(n)]
        names_dead = [n for n in self.shapes if is_dead(n)]
        self.offsets = OrderedDict()
        off = 0
        for n in names_live:
            self.offsets[n] = off
            off += (int(math.prod(self.shapes[n])) + 3) // 4 * 4      # keep every tensor 16-byte aligned
        self.live_numel = off
        for n in names_dead:
            self.offsets[n] = off
            off += (int(math.prod(self.shapes[n])) + 3) // 4 * 4
        self.total_numel = off
        self.device, self.dtype = torch.device(device), dtype
        self.flat = torch.zeros(self.total_numel, dtype=dtype, device=self.device)
        self.views = self._make_views(self.flat)
        # bumped whenever the parameters change (optimiser step, load_state_dict): encoders derive operand formats from the weights
        # (trunk.refresh_weights) and several of them - one per batch size - may share this arena
        self.version = 0

    def _make_views(self, flat):
        v = OrderedDict()
        for n, shape in self.shapes.items():
            o = self.offsets[n]
            v[n] = flat[o:o + int(math.prod(shape))].view(shape)
        return v

    def like(self):
        """A zero arena of the same layout (gradients, Adam slots): returns (flat, views)."""
        flat = torch.zeros_like(self.flat)
        return flat, self._make_views(flat)

    def live(self, flat=None):
        return (self.flat if flat is None else flat)[: self.live_numel]

    def load_state_dict(self, sd, strict=True):
        """Accepts short names ('conv2d/kernel') or full TF names ('Generator/Generator/conv2d/kernel')."""
        for n in self.shapes:
            src = sd.get(n, sd.get(tf_variable_name(self.kind, n)))
            if src is None:
                if strict:
                    raise KeyError("missing parameter %s" % n)
                continue
            src = torch.as_tensor(src)
            if tuple(src.shape) != tuple(self.shapes[n]):
                raise ValueError("shape mismatch for %s: %s vs %s" % (n, tuple(src.shape), self.shapes[n]))
            self.views[n].copy_(src.to(self.dtype))
        self.version += 1

    def state_dict(self, full_names=False):
        return OrderedDict(((tf_variable_name(self.kind, n) if full_names else n), v.detach().clone().cpu())
                           for n, v in self.views.items())

    def live_param_count(self):
        return sum(int(math.prod(s)) for n, s in self.shapes.items() if not is_dead(n))


def init_state_dict(kind, V, S, E=EMBED_DIM, seed=3):
    """Reference initialisers (generator_with_attention.py:21-22; TF defaults for dense/LSTM/LN):
    conv kernels he_normal (truncated normal, sigma = sqrt(2/fan_in)), conv bias 0.05, LN gamma 1 / beta 0,
    dense and LSTM kernels Glorot-uniform, dense bias 0, embedding U(-0.1, 0.1)
    (dataset_creation/map_files_to_triples.py:24).  CPU generator so every rank/path gets identical bits."""
    gen = torch.Generator().manual_seed(seed + (0 if kind == "G" else 1000))
    sd = OrderedDict()
    for name, shape in param_shapes(kind, V, S, E).items():
        leaf = name.split("/")[-1]
        if name.startswith("conv2d") and leaf == "kernel":
            std = math.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
        elif name.startswith("conv2d") and leaf == "bias":
            t = torch.full(shape, 0.05)
        elif leaf == "gamma":
            t = torch.ones(shape)
        elif leaf == "beta":
            t = torch.zeros(shape)
        elif leaf == "kernel":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=gen) * 2 - 1) * lim
        elif leaf == "bias":
            t = torch.zeros(shape)
        elif name == "W":
            t = torch.rand(shape, generator=gen) * 0.2 - 0.1
        else:
            raise KeyError(name)
        sd[name] = t
    return sd
