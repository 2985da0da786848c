"""Independent plain-loop NumPy restatement of the small ops (test infrastructure).  Written from the TF-1.x op
semantics in SURVEY.md Appendix A without looking at oracle/sgg_oracle.py's tensor formulation, so that the two
restatements cross-check each other (the reference itself cannot run here: parity unpinned, see sgg_oracle.py)."""
import math

import numpy as np


def same_pads(n, k, s):
    out = (n + s - 1) // s
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, b, s):
    """x [B,H,W,Ci], w [kh,kw,Ci,Co] (HWIO, cross-correlation), extra SAME pad goes to the bottom/right (A.1)."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Ho, pt, _ = same_pads(H, kh, s)
    Wo, pl, _ = same_pads(W, kw, s)
    y = np.zeros((B, Ho, Wo, Co), dtype=np.float64)
    for bb in range(B):
        for ho in range(Ho):
            for wo in range(Wo):
                for i in range(kh):
                    for j in range(kw):
                        hi, wi = ho * s - pt + i, wo * s - pl + j
                        if 0 <= hi < H and 0 <= wi < W:
                            y[bb, ho, wo, :] += x[bb, hi, wi, :] @ w[i, j]
                y[bb, ho, wo, :] += b
    return y


def layer_norm(x, gamma, beta, eps=1e-12):
    """per sample over all non-batch axes; gamma/beta over the last axis (A.2)."""
    y = np.empty_like(x, dtype=np.float64)
    for bb in range(x.shape[0]):
        v = x[bb].astype(np.float64)
        mean = v.sum() / v.size
        var = ((v - mean) ** 2).sum() / v.size
        y[bb] = (v - mean) / math.sqrt(var + eps) * gamma + beta
    return y


def elu(x):
    return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1.0)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lnlstm_cell(x, c, h, kernel, ln):
    """LayerNormBasicLSTMCell (A.5): ln = list of 5 (gamma, beta) for input, transform, forget, output, state."""
    B, n = c.shape
    new_c, new_h = np.zeros_like(c), np.zeros_like(h)
    for bb in range(B):
        z = np.concatenate([x[bb], h[bb]]) @ kernel
        i, j, f, o = z[0:n], z[n:2 * n], z[2 * n:3 * n], z[3 * n:4 * n]
        i = layer_norm(i[None], *ln[0])[0]
        j = layer_norm(j[None], *ln[1])[0]
        f = layer_norm(f[None], *ln[2])[0]
        o = layer_norm(o[None], *ln[3])[0]
        cc = c[bb] * sigmoid(f + 1.0) + sigmoid(i) * np.tanh(j)
        cc = layer_norm(cc[None], *ln[4])[0]
        new_c[bb] = cc
        new_h[bb] = np.tanh(cc) * sigmoid(o)
    return new_h, new_c


def attention(ctx, c, W, b):
    """ctx [B,L,C]; e = [flatten(ctx), c] @ W + b; softmax over L; weighted sum of rows (A.4)."""
    B, L, C = ctx.shape
    z = np.zeros((B, C))
    alpha = np.zeros((B, L))
    for bb in range(B):
        e = np.concatenate([ctx[bb].reshape(-1), c[bb]]) @ W + b
        e = np.exp(e - e.max())
        alpha[bb] = e / e.sum()
        for l in range(L):
            z[bb] += alpha[bb, l] * ctx[bb, l]
    return z, alpha


def tf_adam(theta, g, m, v, t, lr=1e-4, b1=0.5, b2=0.9, eps=1e-8):
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def argmax_first(x):
    out = np.zeros(x.shape[:-1], dtype=np.int64)
    for idx in np.ndindex(*x.shape[:-1]):
        best, bi = -np.inf, 0
        for k in range(x.shape[-1]):
            if x[idx + (k,)] > best:
                best, bi = x[idx + (k,)], k
        out[idx] = bi
    return out
