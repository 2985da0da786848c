"""WGAN-GP critic step and generator step (one "G+D step" = critic_iters critic updates + one generator update).

Reference: train.py:239-266 (tfgan gan_model / gan_loss with wasserstein losses + one-sided gradient penalty,
two Adam optimisers, variable partition by name prefix) and the loop body train.py:362-368.

Critic step:   G forward (fake logits, constant for D) -> D encoder forward ONCE (fake / real / interpolated share
               the images) -> critic head on the 3B-row super-batch -> first-order backward (parameter gradients
               from the fake and real rows; g = d sum(D(x_hat)) / d x_hat from the interpolated rows) -> penalty
               and v = lambda * dGP/dg -> dual-number head pass on the interpolated rows (JVP along v, then the same
               backward evaluated on duals = gradient of the penalty) -> encoder backward -> [all-reduce] -> TF-Adam.
Generator step: G forward -> D forward with the updated critic -> backward through the critic head to the fake
               logits -> generator head + encoder backward -> [all-reduce] -> TF-Adam.
No autograd tape is used anywhere in this path; every arithmetic op is a HIP kernel behind the C ABI.
"""
from __future__ import annotations

import math

import contextlib

import torch

from .head import Head
from .params import ADAM_B1, ADAM_B2, ADAM_EPS, ADAM_LR, EMBED_DIM, FEAT_C, NUM_UNITS, T_STEPS, ParamArena
from .trunk import Trunk


def tf_adam_lr_t(t, lr=ADAM_LR, b1=ADAM_B1, b2=ADAM_B2):
    """tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)."""
    return lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)


class Network:
    """Parameters (+ gradient / Adam arenas), encoder and head of one network for ONE batch size.

    share=<Network>: the variables of the reference graph exist once however many times build_* runs (tf.AUTO_REUSE, train.py:86,91;
    the graph is batch-dynamic: generator_with_attention.py:74-75 reshape to [-1, ...], train.py:29-30,199-203,297-298 feed B, B/2
    and B/2 x 8 rows through the same ops).  A Network built with `share` uses the other one's parameter, gradient and Adam arenas
    and optimiser step count, and owns only the activation buffers of its own batch size."""

    def __init__(self, K, kind, V, S, B, E=EMBED_DIM, device=None, dtype=torch.float32, state_dict=None, share=None):
        self.K, self.kind = K, kind
        device = device if device is not None else K.device
        if share is not None:
            assert state_dict is None and (share.kind, share.arena.V, share.arena.S, share.arena.E) == (kind, V, S, E)
            self.arena, self.grad_flat, self.grads = share.arena, share.grad_flat, share.grads
            self.m_flat, self.v_flat, self.opt = share.m_flat, share.v_flat, share.opt
        else:
            self.arena = ParamArena(kind, V, S, E, device=device, dtype=dtype)
            if state_dict is not None:
                self.arena.load_state_dict(state_dict)
            self.grad_flat, self.grads = self.arena.like()
            self.m_flat, _ = self.arena.like()
            self.v_flat, _ = self.arena.like()
            # t: Adam step count; pending: in-flight gradient all-reduce (dp.PendingReduce) whose Adam step is deferred
            self.opt = {"t": 0, "pending": None}
        self.trunk = Trunk(K, self.arena, self.grads, B, S)
        self.head = Head(K, kind, self.arena, self.grads, B, self.trunk.L)

    adam_t = property(lambda self: self.opt["t"], lambda self, v: self.opt.__setitem__("t", v))
    pending = property(lambda self: self.opt["pending"], lambda self, v: self.opt.__setitem__("pending", v))

    def zero_grads(self):
        assert self.pending is None
        self.K.fill(self.arena.live(self.grad_flat), 0.0)

    def update(self, reducer):
        """Gradients are complete: start their all-reduce (if data parallel) and defer the Adam step until the
        weights are next needed (finish_update), so the collective overlaps with the other network's encoder."""
        if reducer is None:
            self.adam_step(1.0)
        else:
            self.pending = reducer(self)

    def finish_update(self):
        if self.pending is not None:
            scale = self.pending.wait()
            self.pending = None
            self.adam_step(scale)

    def adam_step(self, grad_scale=1.0):
        self.adam_t += 1
        a = self.arena
        self.K.adam(a.live(), a.live(self.grad_flat), a.live(self.m_flat), a.live(self.v_flat),
                    tf_adam_lr_t(self.adam_t), ADAM_B1, ADAM_B2, ADAM_EPS, grad_scale)
        a.version += 1                       # (encoders of other batch sizes on this arena re-derive their operand formats lazily)
        self.trunk.refresh_weights()


class GanStep:
    """The training hot path for a fixed (B, S, V). `reducer(network)` (optional) all-reduces network.grad_flat
    across data-parallel ranks and returns the scale to apply to the gradient (1/world_size)."""

    def __init__(self, K, V, S, B, lam=10.0, E=EMBED_DIM, g_state=None, d_state=None, dtype=torch.float32, reducer=None,
                 G=None, D=None, overlap_streams=False, head_side_stream=None):
        self.K, self.V, self.S, self.B, self.lam = K, V, S, B, float(lam)
        self.G = G if G is not None else Network(K, "G", V, S, B, E, dtype=dtype, state_dict=g_state)
        self.D = D if D is not None else Network(K, "D", V, S, B, E, dtype=dtype, state_dict=d_state)
        self.reducer = reducer
        dev = self.G.arena.flat.device
        z = lambda *s: torch.zeros(s, device=dev, dtype=dtype)
        self.TRI = z(3 * B, T_STEPS, V)          # rows: fake | real (one-hot) | interpolated
        self.gbuf = z(B, T_STEPS, V)             # g = d sum(D(x_hat)) / d x_hat
        self.vbuf = z(B, T_STEPS, V)             # lambda * dGP/dg
        self.slopes, self.pen = z(B), z(B)
        self.dfake = z(1, B, T_STEPS, V)
        self.d_losses, self.g_losses = z(4), z(4)
        self.tokens = torch.zeros((B, T_STEPS), dtype=torch.int64, device=dev)
        # Optional second HIP stream: the two encoders are independent until the critic head needs the fake triples,
        # so D's encoder forward can run next to G's forward (the HBM-bound LayerNorm passes of one network overlap
        # the MFMA-bound convolutions of the other, launch tails are filled).  Measured +3 % triples/s; off by
        # default because concurrent kernels make per-kernel durations (the roofline measurement) meaningless.
        self._g_reuse, self._g_reuse_armed = None, False       # (images, ctx) of G's encoder within one train_iteration
        # (option side_priority: priority of the side streams - everything on them is off the critical chain of the main stream)
        prio = int(getattr(K, "side_priority", 0))
        self.side = torch.cuda.Stream(device=dev, priority=prio) if (overlap_streams and dev.type == "cuda") else None
        if self.side is not None:
            # backward: filter gradients beside the dgrad -> LayerNorm-backward chain (trunk.enable_wgrad_overlap)
            self.G.trunk.enable_wgrad_overlap(self.side)
            self.D.trunk.enable_wgrad_overlap(self.side)
        # The recurrent heads are chains of short dependent launches.  Their parameter-gradient work (95 of the ~330 launches per step:
        # every weight-gradient GEMM and column sum) is off that chain: with a stream of its own the backward pass DEFERS it there
        # behind one fork per pass (head.py; bit-identical results), and nothing waits for it until head.join() in front of the
        # optimiser - the chain, and the encoder backward after it, no longer carry those launches: 44.32 / 44.32 / 44.39 against
        # 45.12 / 45.01 / 45.20 ms per step (same box, interleaved; profiles/r04_head_deferred_grads_ab.log).  Part of the two-stream
        # schedule by default (head_side_stream=None follows overlap_streams).  Round 3's form - a fork per time step and a join at
        # the end of every pass - measured equal to none.
        if head_side_stream is None:
            head_side_stream = overlap_streams
        self.head_side = torch.cuda.Stream(device=dev, priority=prio) if (head_side_stream and dev.type == "cuda") else None
        self.G.head.enable_side_stream(self.head_side)
        self.D.head.enable_side_stream(self.head_side)

    # ------------------------------------------------------------------------------------------------
    def _g_early_stream(self, images):
        """Option g_early (default on, two-stream schedule): G's encoder forward of THIS update depends on nothing the previous update
        still computes once that update's G head has run (a critic update leaves G's weights alone): it may start on a stream of its
        own right there - beside the critic's heads, which are a chain of short launches that leaves the chip idle, and its encoder
        backward.  Returns that stream (already waiting for the event), or None.  Same kernels, same operands: bit-identical;
        43.17 / 43.15 against 43.60 / 43.73 ms per step (profiles/r04_g_early_ab.log)."""
        ev, ev_images = getattr(self, "_ev_g_free", None), getattr(self, "_ev_g_images", None)
        self._ev_g_free = self._ev_g_images = None
        if self.side is None or ev is None or not getattr(self.K, "g_early", 0) or self.G.pending is not None or self._g_reuse is not None:
            return None
        # only for the minibatch tensor the critic update ran on, unmodified (train.py:175-190 repeats each batch for every update of an
        # iteration): the early stream waits for nothing the main stream enqueued after that update's G head, so a tensor produced there
        # since (another batch, an augmentation in place) would be read too early
        if ev_images is None or ev_images[0] is not images or ev_images[1] != images._version:
            return None
        # ... and only for the parameters that critic update left: a state-dict load or an optimiser step through another batch size's
        # Network on the shared arena since then would make trunk.forward re-derive the weight formats (refresh_weights) on the early
        # stream, unordered against the main stream's writes
        if ev_images[2] != (self.G.arena.version, self.G.adam_t):
            return None
        if getattr(self, "xs", None) is None:
            self.xs = torch.cuda.Stream(device=images.device)
        self.xs.wait_event(ev)
        return self.xs

    def generator_forward(self, images, noise, for_backward=True, early=None):
        """Generator.build_generator: fake logits [B,3,V] (a view of the critic's input slab).

        Inside train_iteration(..., reuse_g_encoder=True) G's ENCODER runs once per iteration: every update of an iteration sees
        the same minibatch (train.py:175-190 repeats each batch CRITIC_ITERS + 1 times) and G's weights only change at its end, so
        the encoder output, the step-invariant attention product and the activations kept for G's backward are those of the first
        call; only the recurrent head (fresh noise) is re-run.  bench.py never does this (every update recomputes everything)."""
        G = self.G
        if self._g_reuse is not None and self._g_reuse[0] is images:
            # the kept encoder output is only valid for the tensor's contents at the first call: an in-place write to the minibatch
            # (augmentation, a loader refilling its device buffer) or an optimiser step of G since then would silently train on stale
            # activations.  (Invariant of the schedule: no G.head.backward runs between the first generator_forward of an iteration
            # and generator_step, so the dP / dctx accumulators head.precompute cleared are still zero there.)
            assert self._g_reuse[2] == (images.data_ptr(), images._version, G.adam_t), \
                "G-encoder reuse: the minibatch tensor was modified in place (or G was updated) inside one iteration"
            ctx = self._g_reuse[1]
        else:
            keep = for_backward or self._g_reuse_armed
            if early is not None:        # (the caller made `early` wait for everything this forward depends on)
                cap = int(getattr(self.K, "g_early_cus", 0))
                kw = {"cu_cap": cap} if cap else {}
                with torch.cuda.stream(early):
                    ctx = G.trunk.forward(images, keep, **kw) if keep is False else G.trunk.forward(images, **kw)
                    G.head.precompute(ctx)
                torch.cuda.current_stream().wait_stream(early)
            else:
                ctx = G.trunk.forward(images, keep) if keep is False else G.trunk.forward(images)
                G.head.precompute(ctx)
            if self._g_reuse_armed:
                self._g_reuse = (images, ctx, (images.data_ptr(), images._version, G.adam_t))
        st = G.head.state(1, self.B)
        G.head.forward(st, ctx, noise)
        return st, ctx

    def _d_encoder_on_side_stream(self, images, zero_grads, for_backward=True):
        """D.finish_update (pending all-reduce + Adam), D's encoder forward and the step-invariant attention product,
        enqueued on the side stream; returns ctx. Call _join_side() before anything on the main stream reads them."""
        D = self.D
        if self.side is None:
            D.finish_update()
            if zero_grads:
                D.zero_grads()
            ctx = D.trunk.forward(images, for_backward) if for_backward is False else D.trunk.forward(images)
            D.head.precompute(ctx)
            return ctx
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        cap = int(getattr(self.K, "d_side_cus", 0))      # (option d_side_cus: as g_early_cus, for D's forward beside G's forward and head)
        kw = {"cu_cap": cap} if cap else {}
        with torch.cuda.stream(self.side):
            D.finish_update()
            if zero_grads:
                D.zero_grads()
            ctx = D.trunk.forward(images, for_backward, **kw) if for_backward is False else D.trunk.forward(images, **kw)
            D.head.precompute(ctx)
        return ctx

    def _join_side(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def critic_step(self, images, labels, noise, alpha):
        """One disc_train_op (train.py:365). labels int64 [B,3]; noise [B,512]; alpha [B]. Returns self.d_losses
        = (disc_cost, wasserstein term, gradient penalty, mean D(fake)) as a device tensor."""
        K, B, V, D = self.K, self.B, self.V, self.D
        fake_rows, real_rows, hat_rows = self.TRI[:B], self.TRI[B:2 * B], self.TRI[2 * B:]
        # Data parallel: the network WITHOUT a gradient all-reduce in flight goes first, so that its encoder forward
        # runs under the other network's collective before anything waits for it (critic_iters = 1: G's reduce from the
        # last generator step hides under D's encoder; critic_iters > 1: D's reduce from the previous critic update
        # hides under G's forward).  With a side stream the wait is enqueued there and never blocks the main stream.
        if D.pending is None or self.side is not None:
            early = self._g_early_stream(images)      # (critic_iters > 1: after another critic update)
            ctx = self._d_encoder_on_side_stream(images, zero_grads=True)
            self.G.finish_update()
            gst, _ = self.generator_forward(images, noise, for_backward=False, early=early)     # the critic update never differentiates G
        else:
            self.G.finish_update()
            gst, _ = self.generator_forward(images, noise, for_backward=False)
            ctx = self._d_encoder_on_side_stream(images, zero_grads=True)
        if self.side is not None and getattr(K, "g_early", 0):
            # G's encoder buffers are free from here on, and this critic update does not touch G's weights (_g_early_stream)
            self._ev_g_free = torch.cuda.Event()
            self._ev_g_free.record()
            self._ev_g_images = (images, images._version, (self.G.arena.version, self.G.adam_t))
        self._join_side()
        fake_rows.copy_(gst.OUT[0])
        K.onehot(labels, real_rows)
        K.interpolate(real_rows, fake_rows, alpha, hat_rows)
        # ---- first-order pass on the 3B-row super-batch ------------------------------------------------
        st = D.head.state(1, 3 * B)
        D.head.forward(st, ctx, [self.TRI], labels, (B, 2 * B))
        inv = 1.0 / (B * T_STEPS)
        K.fill(st.dOUT[0][:B], inv)               # d mean(D(fake))
        K.fill(st.dOUT[0][B:2 * B], -inv)         # d -mean(D(real))
        K.fill(st.dOUT[0][2 * B:], 1.0)           # d sum(D(x_hat)) -> g
        D.head.backward(st, ctx, [self.TRI], R_w=2 * B, labels=labels, label_rows=(B, 2 * B))
        ind = D.head.in_dim
        for t in range(T_STEPS):
            K.gemm_nt(st.dXH[t][0][2 * B:, FEAT_C:ind], D.head.W_emb, self.gbuf[:, t, :])
        K.gp_fwd(self.gbuf, self.slopes, self.pen)
        K.gp_bwd(self.gbuf, self.slopes, self.pen, self.vbuf, self.lam)
        K.wgan_losses(st.OUT[0].view(3 * B, T_STEPS), self.pen, self.lam, B, T_STEPS, True, self.d_losses)
        # ---- gradient of lambda*GP: dual-number pass on the interpolated rows ------------------------------
        st2 = D.head.state(2, B)
        u2 = [hat_rows, self.vbuf]
        D.head.forward(st2, ctx, u2)
        K.fill(st2.dOUT[0], 1.0)                  # cotangent of the tangent output (= d(lambda*GP)/d JVP)
        K.fill(st2.dOUT[1], 0.0)
        D.head.backward(st2, ctx, u2, R_w=B)
        # W enters g = delta_e @ W^T directly as well: handled by the tangent input v @ W above (u2[1])
        dctx = D.head.finish_backward(ctx)
        D.trunk.backward(dctx)
        D.head.join()
        D.update(self.reducer)
        return self.d_losses

    def critic_loss(self, images, labels, noise, alpha, out=None):
        """disc_cost of a minibatch WITHOUT an update: what `sess.run(self.disc_cost, feed_dict = {handle: val_handle})` evaluates
        for the validation-loss early stop (train.py:375-377).  Same forward as critic_step; the head backward runs only as far as
        g = d sum(D(x_hat)) / d x_hat needs it (no parameter gradient is touched, R_w = 0), no encoder backward, no Adam.
        Returns (disc_cost, wasserstein term, gradient penalty, mean D(fake)) in `out` (default: a buffer of its own)."""
        K, B, V, D = self.K, self.B, self.V, self.D
        assert self._g_reuse is None and not self._g_reuse_armed, "critic_loss inside an iteration with G-encoder reuse"
        if out is None:
            if getattr(self, "val_losses", None) is None:
                self.val_losses = torch.zeros_like(self.d_losses)
            out = self.val_losses
        fake_rows, real_rows, hat_rows = self.TRI[:B], self.TRI[B:2 * B], self.TRI[2 * B:]
        self.flush()
        ctx = D.trunk.forward(images, False)
        D.head.precompute(ctx)
        gst, _ = self.generator_forward(images, noise, for_backward=False)
        fake_rows.copy_(gst.OUT[0])
        K.onehot(labels, real_rows)
        K.interpolate(real_rows, fake_rows, alpha, hat_rows)
        st = D.head.state(1, 3 * B)
        D.head.forward(st, ctx, [self.TRI], labels, (B, 2 * B))
        K.fill(st.dOUT[0][:2 * B], 0.0)
        K.fill(st.dOUT[0][2 * B:], 1.0)           # d sum(D(x_hat)) -> g
        D.head.backward(st, ctx, [self.TRI], R_w=0)
        ind = D.head.in_dim
        for t in range(T_STEPS):
            K.gemm_nt(st.dXH[t][0][2 * B:, FEAT_C:ind], D.head.W_emb, self.gbuf[:, t, :])
        K.gp_fwd(self.gbuf, self.slopes, self.pen)
        K.wgan_losses(st.OUT[0].view(3 * B, T_STEPS), self.pen, self.lam, B, T_STEPS, True, out)
        return out

    def generator_step(self, images, noise):
        """One gen_train_op (train.py:368). Returns self.g_losses; g_losses[3] = mean D(fake) = -gen_cost."""
        K, B, G, D = self.K, self.B, self.G, self.D
        G.finish_update()
        G.zero_grads()
        if D.pending is None or self.side is not None:
            early = self._g_early_stream(images)
            ctx = self._d_encoder_on_side_stream(images, zero_grads=False, for_backward=False)   # independent of G's forward
            gst, gctx = self.generator_forward(images, noise, early=early)
        else:
            # the critic-gradient all-reduce launched at the end of critic_step runs under G's forward; only then does
            # D.finish_update() wait for it
            gst, gctx = self.generator_forward(images, noise)
            ctx = self._d_encoder_on_side_stream(images, zero_grads=False, for_backward=False)   # only D's head is differentiated here
        fake = gst.OUT[0]
        self._join_side()
        st = D.head.state(1, B, "g")
        D.head.forward(st, ctx, [fake])
        K.wgan_losses(st.OUT[0].view(B, T_STEPS), None, 0.0, B, T_STEPS, False, self.g_losses)
        K.fill(st.dOUT[0], -1.0 / (B * T_STEPS))  # gen_cost = -mean(D(G(x)))
        D.head.backward(st, ctx, [fake], R_w=0)
        ind = D.head.in_dim
        for t in range(T_STEPS):
            K.gemm_nt(st.dXH[t][0][:, FEAT_C:ind], D.head.W_emb, gst.dOUT[0][:, t, :])
        G.head.backward(gst, gctx, None, R_w=B)
        dctx = G.head.finish_backward(gctx)
        G.trunk.backward(dctx)
        G.head.join()
        G.update(self.reducer)
        return self.g_losses

    def flush(self):
        """Apply any deferred optimiser update (before reading weights / at the end of the timed region)."""
        self.D.finish_update()
        self.G.finish_update()

    def train_iteration(self, images, labels, noises, alphas, critic_iters=1, reuse_g_encoder=False):
        """Loop body of train.py:362-368: critic_iters critic updates then one generator update on one minibatch,
        fresh noise / alpha per update.  reuse_g_encoder: G's encoder forward once per iteration (generator_forward)."""
        with self.iteration(reuse_g_encoder):
            for i in range(critic_iters):
                self.critic_step(images, labels, noises[i], alphas[i])
            self.generator_step(images, noises[critic_iters])

    @contextlib.contextmanager
    def iteration(self, reuse_g_encoder=False):
        """The updates of ONE minibatch (train.py:362-368).  reuse_g_encoder: see generator_forward."""
        self._g_reuse, self._g_reuse_armed = None, bool(reuse_g_encoder)
        try:
            yield self
        finally:
            self._g_reuse, self._g_reuse_armed = None, False

    def argmax_tokens(self, logits):
        """tf.argmax(fake_inputs, -1) (train.py:270)."""
        self.K.argmax_rows(logits, self.tokens.view(-1))
        return self.tokens
