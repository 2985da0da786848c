"""Kernel-level CPU reference (test infrastructure, NOT product code).

`RefKernels` restates, with plain torch CPU ops, the contract of every entry point of the C ABI at the same
tensor-level interface as `sgg_amd.lib.HipKernels`.  Uses:
  * tests -m gpu : each HIP kernel is compared with the matching method here on the same inputs;
  * tests (CPU)  : the host orchestration (sgg_amd.trunk/head/step) is run with this object injected in place
                   of HipKernels and compared with the autograd oracle (oracle/sgg_oracle.py), which pins the
                   hand-written backward schedule and the dual-number treatment of the gradient penalty.
First-order backward formulas are obtained from torch.autograd / torch.func on the oracle's forward
definitions; the dual ("bwd2") variants by torch.func.jvp over those — independent of the hand-derived
device code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import torch
from torch.func import jvp, vjp, vmap

from . import sgg_oracle as O


def _rowmap(R, B, device=None):
    return torch.arange(R, device=device) % B


class RefKernels:
    name = "ref"

    def __init__(self, device="cpu"):
        self.device = torch.device(device)

    # -- conv encoder ----------------------------------------------------------------------------------
    def hwio_to_hwoi(self, w, wt):
        wt.copy_(w.permute(0, 1, 3, 2))

    def conv_fwd(self, x, w_hwio, w_fwd, bias, y, stride):
        y.copy_(O.conv2d_same(x, w_hwio, bias, stride))

    def conv_dgrad(self, dy, w_hwio, dx, stride):
        x0 = torch.zeros_like(dx, requires_grad=True)
        y = O.conv2d_same(x0, w_hwio, torch.zeros(w_hwio.shape[3], dtype=dx.dtype), stride)
        (g,) = torch.autograd.grad(y, x0, dy)
        dx.copy_(g)

    def conv_wgrad(self, x, dy, dw, stride):
        w0 = torch.zeros_like(dw, requires_grad=True)
        y = O.conv2d_same(x, w0, torch.zeros(dw.shape[3], dtype=dw.dtype), stride)
        (g,) = torch.autograd.grad(y, w0, dy)
        dw.copy_(g)

    def ln_elu_fwd(self, y, gamma, beta, a, stats, region=None):
        if region is not None:      # valid window of a canvas (sgg_hip.h): normalise the window, zeros elsewhere
            r0, c0, hv, wv = region
            a.zero_()
            y, a = y[:, r0:r0 + hv, c0:c0 + wv], a[:, r0:r0 + hv, c0:c0 + wv]
        a.copy_(O.elu(O.layer_norm_tf(y, gamma, beta)))
        mean = y.mean(dim=(1, 2, 3))
        var = ((y - mean[:, None, None, None]) ** 2).mean(dim=(1, 2, 3))
        stats[:, 0] = mean
        stats[:, 1] = torch.rsqrt(var + O.LN_EPS)

    def ln_elu_bwd(self, y, da, gamma, beta, stats, dy, dgamma, dbeta, dbias_prev, region=None):
        if region is not None:
            r0, c0, hv, wv = region
            dy.zero_()
            y, da, dy = (t[:, r0:r0 + hv, c0:c0 + wv] for t in (y, da, dy))
        y0 = y.detach().clone().requires_grad_(True)
        g0 = gamma.detach().clone().requires_grad_(True)
        b0 = beta.detach().clone().requires_grad_(True)
        a = O.elu(O.layer_norm_tf(y0, g0, b0))
        gy, gg, gb = torch.autograd.grad(a, (y0, g0, b0), da)
        dy.copy_(gy)
        dgamma.copy_(gg)
        dbeta.copy_(gb)
        if dbias_prev is not None:
            dbias_prev.copy_(gy.sum(dim=(0, 1, 2)))

    # -- heads -----------------------------------------------------------------------------------------
    def spatial_mean_fwd(self, ctx, out_c, out_h):
        R = out_c.shape[0]
        m = ctx.mean(dim=1)[_rowmap(R, ctx.shape[0])]
        out_c.copy_(m)
        out_h.copy_(m)

    def spatial_mean_bwd(self, dc0, dh0, dctx, accumulate):
        B, L, C = dctx.shape
        R = dc0.shape[0]
        s = (dc0 + dh0).reshape(R // B, B, C).sum(dim=0) / L
        upd = s[:, None, :].expand(B, L, C)
        if accumulate:
            dctx.add_(upd)
        else:
            dctx.copy_(upd)

    def gemm_nn(self, A, Bm, C, bias=None, accumulate=False):
        r = A @ Bm
        if bias is not None:
            r = r + bias
        C.copy_(C + r if accumulate else r)

    def gemm_nt(self, A, Bm, C, accumulate=False):
        r = A @ Bm.t()
        C.copy_(C + r if accumulate else r)

    def gemm_tn(self, A, Bm, C, accumulate=False):
        r = A.t() @ Bm
        C.copy_(C + r if accumulate else r)

    def attn_ctx_fwd(self, ctx_flat, W_ctx, bias, P):
        P.copy_(ctx_flat @ W_ctx + bias)

    def attn_ctx_dgrad(self, dP, W_ctx, dctx_flat, accumulate=True):
        r = dP @ W_ctx.t()
        dctx_flat.copy_(dctx_flat + r if accumulate else r)

    def attn_ctx_wgrad(self, ctx_flat, dP, dW_ctx, accumulate=True):
        r = ctx_flat.t() @ dP
        dW_ctx.copy_(dW_ctx + r if accumulate else r)

    def attn_step_fwd(self, P, ec, ctx, alpha, z):
        B, L, C = ctx.shape
        R = ec.shape[1]
        rm = _rowmap(R, B)
        Pr, ctxr = P[rm], ctx[rm]

        def f(ec_):
            al = torch.softmax(Pr + ec_, dim=1)
            return al, torch.einsum("rl,rlc->rc", al, ctxr)

        if ec.shape[0] == 1:
            al, zz = f(ec[0])
            alpha[0].copy_(al)
            z[0].copy_(zz)
        else:
            (al, zz), (ald, zd) = jvp(f, (ec[0],), (ec[1],))
            alpha[0].copy_(al); alpha[1].copy_(ald)
            z[0].copy_(zz); z[1].copy_(zd)

    def attn_step_bwd(self, ctx, alpha, dz, de, dP, dctx, accumulate):
        B, L, C = ctx.shape
        R = alpha.shape[1]
        rm = _rowmap(R, B)
        ctxr = ctx[rm]

        def bw(al, dz_):
            dal = torch.einsum("rc,rlc->rl", dz_, ctxr)
            de_ = al * (dal - (dal * al).sum(dim=1, keepdim=True))
            dctx_rows = al[:, :, None] * dz_[:, None, :]
            return de_, dctx_rows

        if alpha.shape[0] == 1:
            de_, dctx_rows = bw(alpha[0], dz[0])
            de[0].copy_(de_)
            pc_de, pc_ctx = de_, dctx_rows
        else:
            (de_r, _), (de_d, dctx_d) = jvp(bw, (alpha[0], dz[0]), (alpha[1], dz[1]))
            de[0].copy_(de_r); de[1].copy_(de_d)
            pc_de, pc_ctx = de_d, dctx_d
        upd_ctx = pc_ctx.reshape(R // B, B, L, C).sum(dim=0)
        upd_P = pc_de.reshape(R // B, B, L).sum(dim=0)
        if accumulate:
            dctx.add_(upd_ctx); dP.add_(upd_P)
        else:
            dctx.copy_(upd_ctx); dP.copy_(upd_P)

    @staticmethod
    def _cell(g, c, ln):
        """pointwise part of LayerNormBasicLSTMCell on one or many rows; ln [10,512]."""
        i, j, f, o = torch.chunk(g, 4, dim=-1)
        i = O.layer_norm_tf(i.unsqueeze(0), ln[0], ln[1])[0] if g.dim() == 1 else O.layer_norm_tf(i, ln[0], ln[1])
        j = O.layer_norm_tf(j.unsqueeze(0), ln[2], ln[3])[0] if g.dim() == 1 else O.layer_norm_tf(j, ln[2], ln[3])
        f = O.layer_norm_tf(f.unsqueeze(0), ln[4], ln[5])[0] if g.dim() == 1 else O.layer_norm_tf(f, ln[4], ln[5])
        o = O.layer_norm_tf(o.unsqueeze(0), ln[6], ln[7])[0] if g.dim() == 1 else O.layer_norm_tf(o, ln[6], ln[7])
        cp = c * torch.sigmoid(f + O.FORGET_BIAS) + torch.sigmoid(i) * torch.tanh(j)
        cn = O.layer_norm_tf(cp.unsqueeze(0), ln[8], ln[9])[0] if g.dim() == 1 else O.layer_norm_tf(cp, ln[8], ln[9])
        h = torch.tanh(cn) * torch.sigmoid(o)
        return cn, h

    def lstm_fwd(self, gates, c_prev, ln_params, c_new, h_new):
        f = lambda g, c: self._cell(g, c, ln_params)
        if gates.shape[0] == 1:
            cn, h = f(gates[0], c_prev[0])
            c_new[0].copy_(cn); h_new[0].copy_(h)
        else:
            (cn, h), (cnd, hd) = jvp(f, (gates[0], c_prev[0]), (gates[1], c_prev[1]))
            c_new[0].copy_(cn); c_new[1].copy_(cnd)
            h_new[0].copy_(h); h_new[1].copy_(hd)

    def lstm_bwd(self, gates, c_prev, ln_params, dh, dc_new, dgates, dc_prev, pgrad):
        np_ = gates.shape[0]
        if dc_new is None:
            dc_new = torch.zeros_like(c_prev)

        def row_bw(g, c, dh_, dcn_):
            _, fn = vjp(lambda g_, c_, ln_: self._cell(g_, c_, ln_), g, c, ln_params)
            return fn((dcn_, dh_))          # (dg, dc, dln[10,512])

        bw = vmap(row_bw)
        if np_ == 1:
            dg, dc, dln = bw(gates[0], c_prev[0], dh[0], dc_new[0])
            dgates[0].copy_(dg); dc_prev[0].copy_(dc); pgrad.copy_(dln)
        else:
            (dg, dc, _), (dgd, dcd, dlnd) = jvp(bw, (gates[0], c_prev[0], dh[0], dc_new[0]),
                                               (gates[1], c_prev[1], dh[1], dc_new[1]))
            dgates[0].copy_(dg); dgates[1].copy_(dgd)
            dc_prev[0].copy_(dc); dc_prev[1].copy_(dcd)
            pgrad.copy_(dlnd)

    def colsum(self, X, out, accumulate=False):
        s = X.sum(dim=0)
        out.copy_(out + s if accumulate else s)

    # -- loss / optimiser / misc -----------------------------------------------------------------------
    def onehot(self, labels, out):
        out.copy_(torch.nn.functional.one_hot(labels, out.shape[-1]).to(out.dtype))

    def embed_gather_fwd(self, labels, W, out):
        """tf.matmul(one_hot(labels), W) (discriminator_with_attention.py:86-87 on train.py:173's one-hots)."""
        out.copy_(torch.nn.functional.one_hot(labels, W.shape[0]).to(W.dtype) @ W)

    def embed_gather_bwd(self, labels, dY, dW):
        dW.add_(torch.nn.functional.one_hot(labels, dW.shape[0]).to(dW.dtype).t() @ dY)

    def interpolate(self, real, fake, alpha, out):
        B = real.shape[0]
        a = alpha.reshape(B, *([1] * (real.dim() - 1)))
        out.copy_(real + a * (fake - real))

    def gp_fwd(self, g, slopes, pen):
        B = g.shape[0]
        s = torch.sqrt((g.reshape(B, -1) ** 2).sum(dim=1) + O.GP_EPS)
        slopes.copy_(s)
        pen.copy_(torch.clamp(s - 1.0, min=0.0))

    def gp_bwd(self, g, slopes, pen, v, scale):
        B = g.shape[0]
        coef = scale * (2.0 / B) * pen / slopes
        v.copy_(coef.reshape(B, *([1] * (g.dim() - 1))) * g)

    def wgan_losses(self, d_out, pen, lam, B, T, has_real, out4):
        flat = d_out.reshape(-1)
        mf = flat[: B * T].mean()
        mr = flat[B * T: 2 * B * T].mean() if has_real else torch.zeros((), dtype=d_out.dtype)
        gp = (pen ** 2).mean() if pen is not None else torch.zeros((), dtype=d_out.dtype)
        out4[0] = (mf - mr) + lam * gp
        out4[1] = mf - mr
        out4[2] = gp
        out4[3] = mf

    def adam(self, params, grads, m, v, lr_t, b1, b2, eps, grad_scale=1.0):
        g = grads * grad_scale
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        params.sub_(lr_t * m / (v.sqrt() + eps))

    def argmax_rows(self, x, out):
        out.copy_(O.argmax_tokens(x.reshape(-1, x.shape[-1])))

    def fill(self, t, value):
        t.fill_(value)
