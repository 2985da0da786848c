"""Attention + LayerNorm-LSTM head of both networks: forward, hand-scheduled backward, and their dual-number
evaluation for the gradient-penalty term.

Reference: architectures/generator_with_attention.py:13-18, 74-91 (generator head),
architectures/discriminator_with_attention.py:13-18, 73-93 (critic head); gradients as produced by
optimizer.minimize / tfgan's gradient penalty (train.py:245-250, 265-266).

Design notes (MI355X-first, not a translation of the TF graph):
  * the step-invariant product ctx_flat @ W_ctx (+ bias) of the attention perceptron is computed once per
    forward (`precompute`), the reference graph rebuilds it four times (generator_with_attention.py:81,86);
  * the critic's fake / real / interpolated passes share one feature map, so they run as ONE super-batch of
    R = 3B rows through every GEMM (weights stream once), row r using image r % B;
  * concat([z_hat, u_t, h]) is never materialised: the attention kernel, the embedding GEMM and the gate kernel
    write straight into column ranges of one [R, in+512] buffer that feeds the gate GEMM;
  * tensors carry a leading plane dimension np: 1 = fp32, 2 = dual numbers (real, dual) - the same schedule
    then computes the JVP / second-order backward of the gradient penalty (csrc/dual.h).
"""
from __future__ import annotations

import torch

from .params import FEAT_C, LSTM_LN_SCOPES, NUM_UNITS, T_STEPS

C = FEAT_C
H = NUM_UNITS


def flat2(t):
    """[np, R, w] view whose planes are R rows apart -> [np*R, w] (stacked rows for one GEMM)."""
    np_, R, w = t.shape
    if np_ == 1:
        return t[0]
    assert t.stride(0) == R * t.stride(1) and t.stride(2) == 1
    return t.as_strided((np_ * R, w), (t.stride(1), 1), t.storage_offset())


class HeadState:
    """Activation / cotangent buffers of one head pass with R rows and np planes."""

    def __init__(self, head, np_, R):
        dev, dt = head.device, head.dtype
        z = lambda *s: torch.zeros(s, device=dev, dtype=dt)
        self.np, self.R = np_, R
        W, L, Vout = head.width, head.L, head.Vout
        self.C = [z(np_, R, H) for _ in range(T_STEPS + 1)]
        self.XH = [z(np_, R, W) for _ in range(T_STEPS + 1)]
        self.EC = [z(np_, R, L) for _ in range(T_STEPS)]
        self.AL = [z(np_, R, L) for _ in range(T_STEPS)]
        self.G = [z(np_, R, 4 * H) for _ in range(T_STEPS)]
        self.OUT = z(np_, R, T_STEPS, Vout)
        self.dOUT = z(np_, R, T_STEPS, Vout)
        self.dXH = [z(np_, R, W) for _ in range(T_STEPS + 1)]
        self.dC = [z(np_, R, H) for _ in range(T_STEPS + 1)]
        # (one buffer per time step: the filter-gradient GEMMs that read them may run on the side stream while the chain moves on)
        self.dG = [z(np_, R, 4 * H) for _ in range(T_STEPS)]
        self.dE = [z(np_, R, L) for _ in range(T_STEPS)]
        self.pgrad = z(T_STEPS * R, 10, H)


class Head:
    def __init__(self, K, kind, arena, grad_views, B, L):
        self.K, self.kind, self.B, self.L = K, kind, B, L
        self.device, self.dtype = arena.flat.device, arena.flat.dtype
        p, g = arena.views, grad_views
        self.V, self.E = arena.V, arena.E
        self.in_dim = C + (H if kind == "G" else self.E)
        self.width = self.in_dim + H
        self.Vout = self.V if kind == "G" else 1
        LC = L * C
        self.W_ctx, self.W_c = p["attention_perceptron/kernel"][:LC], p["attention_perceptron/kernel"][LC:]
        self.gW_ctx, self.gW_c = g["attention_perceptron/kernel"][:LC], g["attention_perceptron/kernel"][LC:]
        self.b_att, self.gb_att = p["attention_perceptron/bias"], g["attention_perceptron/bias"]
        self.Kk, self.gKk = p["layer_norm_basic_lstm_cell/kernel"], g["layer_norm_basic_lstm_cell/kernel"]
        first = "layer_norm_basic_lstm_cell/%s/gamma" % LSTM_LN_SCOPES[0]
        o = arena.offsets[first]
        # the ten LN vectors (gamma, beta of input, transform, forget, output, state) are contiguous in the arena
        self.ln = arena.flat[o:o + 10 * H].view(10, H)
        gflat = g[first]
        self.gln = gflat.as_strided((10 * H,), (1,), gflat.storage_offset())
        assert self.ln[2].data_ptr() == p["layer_norm_basic_lstm_cell/transform/gamma"].data_ptr()
        assert self.ln[9].data_ptr() == p["layer_norm_basic_lstm_cell/state/beta"].data_ptr()
        self.W_dec, self.gW_dec = p["decoder/kernel"], g["decoder/kernel"]
        self.b_dec, self.gb_dec = p["decoder/bias"], g["decoder/bias"]
        if kind == "D":
            self.W_emb, self.gW_emb = p["W"], g["W"]
        z = lambda *s: torch.zeros(s, device=self.device, dtype=self.dtype)
        # Second HIP stream for everything OFF the recurrent dependency chain (set by enable_side_stream): the chain of a head pass
        # is  score GEMM -> attention -> gate GEMM -> LSTM pointwise  per step forward and  decoder dgrad -> LSTM backward -> gate
        # dgrad -> attention backward -> score dgrad  per step backward - a few short dependent launches each; the embedding and
        # decoder products of the forward pass and every parameter-gradient GEMM / column sum of the backward pass hang off it and
        # only have to be complete at the end of the pass.  On one stream they sit between the chain's launches (~330 dependent
        # launches of 5-15 us per G+D step).  With a side stream the backward pass DEFERS its parameter-gradient work (95 of those
        # launches) to it behind one fork per pass (see backward); nothing waits for it until join().  Same kernels, same operands,
        # same accumulation order (one side stream, program order) -> bit-identical results.
        self.side = None
        self.P = z(B, L)
        self.dP = z(B, L)
        self.dctx = z(B, L, C)
        self._trashP, self._trashCtx = z(B, L), z(B, L, C)
        self._states = {}

    def enable_side_stream(self, stream):
        self.side = stream

    def _fork(self):
        """Side stream (or None): everything enqueued so far on the current stream is visible to what is enqueued on it next."""
        if self.side is None:
            return None
        self.side.wait_stream(torch.cuda.current_stream())
        return self.side

    def join(self):
        """The current stream waits for the side stream (end of a pass / before the gradients are read)."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def state(self, np_, R, tag=""):
        key = (np_, R, tag)
        if key not in self._states:
            self._states[key] = HeadState(self, np_, R)
        return self._states[key]

    # ------------------------------------------------------------------------------------------------
    def precompute(self, ctx):
        """P = ctx_flat @ W_ctx + b_att, once per feature map; also clears the dP / dctx accumulators."""
        K = self.K
        K.attn_ctx_fwd(ctx.view(self.B, self.L * C), self.W_ctx, self.b_att, self.P)
        K.fill(self.dP, 0.0)
        K.fill(self.dctx, 0.0)

    def forward(self, st, ctx, u, labels=None, label_rows=None):
        """u: generator: noise [R, 512] (reused at t = 0,1,2, generator_with_attention.py:81,86);
        critic: list over planes of triples [R, 3, V] (float one-hot / logits / tangent direction).
        labels int64 [n, 3] + label_rows (lo, hi): rows [lo, hi) of u[0] are the one-hots of `labels` (the real triples,
        train.py:173) - their embedding product tf.matmul(indices, W) (discriminator_with_attention.py:87) is a row gather.
        Fills st.OUT [np, R, 3, Vout]."""
        K, np_, R, ind = self.K, st.np, st.R, self.in_dim
        # (the forward pass runs on ONE stream: its off-chain products - the inputs u_t, the decoder - are read right after the pass,
        #  and a join here would also wait for the parameter-gradient work an earlier pass left on the side stream)
        # ---- known before the loop: the inputs u_t of all three steps (noise copies / embedding products) ------------------------
        for t in range(T_STEPS):
            if self.kind == "G":
                st.XH[t][0][:, C:ind].copy_(u)
                continue
            for pl in range(np_):
                if labels is not None and pl == 0:
                    lo, hi = label_rows
                    for a, b in ((0, lo), (hi, R)):
                        if b > a:
                            K.gemm_nn(u[0][a:b, t, :], self.W_emb, st.XH[t][0][a:b, C:ind])
                    K.embed_gather_fwd(labels[:, t], self.W_emb, st.XH[t][0][lo:hi, C:ind])
                else:
                    K.gemm_nn(u[pl][:, t, :], self.W_emb, st.XH[t][pl][:, C:ind])
        K.spatial_mean_fwd(ctx, st.C[0][0], st.XH[0][0][:, ind:])       # plane 1 (tangent of c0 = h0) stays zero
        for t in range(T_STEPS):
            K.gemm_nn(flat2(st.C[t]), self.W_c, flat2(st.EC[t]))
            K.attn_step_fwd(self.P, st.EC[t], ctx, st.AL[t], st.XH[t][:, :, :C])
            K.gemm_nn(flat2(st.XH[t]), self.Kk, flat2(st.G[t]))
            K.lstm_fwd(st.G[t], st.C[t], self.ln, st.C[t + 1], st.XH[t + 1][:, :, ind:])
            # the decoder (its output is read after the loop)
            K.gemm_nn(st.XH[t + 1][0][:, ind:], self.W_dec, st.OUT[0][:, t, :], self.b_dec)
            if np_ == 2:
                K.gemm_nn(st.XH[t + 1][1][:, ind:], self.W_dec, st.OUT[1][:, t, :])
        return st.OUT

    # ------------------------------------------------------------------------------------------------
    def _wgrad(self, X, dY, dW, R_w):
        """dW += pcot(X^T dY) over the first R_w rows. X, dY: [np, R, *] tensors or per-plane lists."""
        if R_w == 0:
            return
        K = self.K
        if len(X) == 1:
            K.gemm_tn(X[0][:R_w], dY[0][:R_w], dW, accumulate=True)
        else:
            K.gemm_tn(X[0][:R_w], dY[1][:R_w], dW, accumulate=True)
            K.gemm_tn(X[1][:R_w], dY[0][:R_w], dW, accumulate=True)

    def backward(self, st, ctx, u, R_w, labels=None, label_rows=None):
        """Backward of `forward` from st.dOUT. Rows [0, R_w) contribute to parameter / dP / dctx gradients (all
        accumulated); every row gets its data cotangents (st.dXH[t][:, :, 512:in] = cotangent of u_t).
        labels / label_rows as in forward: the embedding gradient of the one-hot rows is a row scatter-add."""
        K, np_, R, ind = self.K, st.np, st.R, self.in_dim
        import contextlib
        on = lambda strm: torch.cuda.stream(strm) if strm is not None else contextlib.nullcontext()
        pc = np_ - 1                                    # plane holding cotangents of real quantities
        assert R_w % self.B == 0 and (np_ == 1 or R_w == R)
        # Everything OFF the chain (every parameter-gradient GEMM and column sum) reads buffers that stay untouched until the next
        # pass on this state.  Without a side stream it runs where it stands; with one it is DEFERRED: collected here, enqueued on the
        # side stream behind ONE fork at the end of the pass, and joined only before the gradients are read (GanStep: head.join()
        # before the optimiser) - the chain, and after it the encoder backward, never wait for it.  Same kernels, same operands, same
        # accumulation order per gradient tensor (one side stream, program order): bit-identical.
        deferred = []
        off = (lambda fn: deferred.append(fn)) if self.side is not None else (lambda fn: fn())
        # ---- off the chain, known before the loop: the decoder's parameter gradients (operands: the forward's h_t and dOUT) --------
        if R_w:
            def dec_grads():
                for t in range(T_STEPS - 1, -1, -1):
                    dout = st.dOUT[:, :, t, :]
                    self._wgrad(st.XH[t + 1][:, :, ind:], dout, self.gW_dec, R_w)
                    K.colsum(dout[pc][:R_w], self.gb_dec, True)
            off(dec_grads)
        for t in range(T_STEPS - 1, -1, -1):
            dout = st.dOUT[:, :, t, :]
            dh = st.dXH[t + 1][:, :, ind:]
            dG, dE = st.dG[t], st.dE[t]
            for pl in range(np_):
                K.gemm_nt(dout[pl], self.W_dec, dh[pl], accumulate=(t < T_STEPS - 1))
            K.lstm_bwd(st.G[t], st.C[t], self.ln, dh, st.dC[t + 1] if t < T_STEPS - 1 else None, dG, st.dC[t],
                       st.pgrad[t * R:(t + 1) * R])
            K.gemm_nt(flat2(dG), self.Kk, flat2(st.dXH[t]))
            dz = st.dXH[t][:, :, :C]
            if R_w:
                K.attn_step_bwd(ctx, st.AL[t][:, :R_w], dz[:, :R_w], dE[:, :R_w], self.dP, self.dctx, True)
            if R_w < R:
                K.attn_step_bwd(ctx, st.AL[t][:, R_w:], dz[:, R_w:], dE[:, R_w:], self._trashP, self._trashCtx, False)
            K.gemm_nt(flat2(dE), self.W_c, flat2(st.dC[t]), accumulate=True)
            # ---- off the chain: the parameter gradients of step t (gate kernel, embedding, score weights), in the serial order ------
            if R_w:
                def step_grads(t=t, dG=dG, dE=dE):
                    self._wgrad(st.XH[t], dG, self.gKk, R_w)
                    if self.kind == "D":
                        if labels is not None and np_ == 1:
                            lo, hi = label_rows
                            assert hi <= R_w
                            for a, b in ((0, lo), (hi, R_w)):
                                if b > a:
                                    K.gemm_tn(u[0][a:b, t, :], st.dXH[t][0][a:b, C:ind], self.gW_emb, accumulate=True)
                            K.embed_gather_bwd(labels[:, t], st.dXH[t][0][lo:hi, C:ind], self.gW_emb)
                        else:
                            self._wgrad([x[:, t, :] for x in u], st.dXH[t][:, :, C:ind], self.gW_emb, R_w)
                    self._wgrad(st.C[t], dE, self.gW_c, R_w)
                off(step_grads)
        if R_w:
            K.spatial_mean_bwd(st.dC[0][pc][:R_w], st.dXH[0][pc][:R_w, ind:], self.dctx, True)

            def ln_grads():
                for t in range(T_STEPS):
                    K.colsum(st.pgrad[t * R:t * R + R_w].view(R_w, 10 * H), self.gln, True)
            off(ln_grads)
        if deferred:
            with on(self._fork()):
                for fn in deferred:
                    fn()

    def finish_backward(self, ctx):
        """Gradients that flow through the step-invariant score P (after every head pass of the step)."""
        K, B, L = self.K, self.B, self.L
        ctx_flat = ctx.view(B, L * C)
        # the parameter gradients (79 MB of attention weights) beside the dgrad that the encoder backward waits for; joined by
        # join() before the gradients are read (GanStep: before the all-reduce / Adam step)
        side = self._fork()
        if side is not None:
            with torch.cuda.stream(side):
                K.colsum(self.dP, self.gb_att, True)
                K.attn_ctx_wgrad(ctx_flat, self.dP, self.gW_ctx, accumulate=True)
        else:
            K.colsum(self.dP, self.gb_att, True)
            K.attn_ctx_wgrad(ctx_flat, self.dP, self.gW_ctx, accumulate=True)
        K.attn_ctx_dgrad(self.dP, self.W_ctx, self.dctx.view(B, L * C), accumulate=True)
        return self.dctx
