"""CPU: the hand-scheduled forward/backward of sgg_amd (trunk.py, head.py, step.py) with the kernel-level
reference injected in place of the HIP binding, against the autograd oracle (oracle/sgg_oracle.py) in fp64.

This pins (a) the backward schedule (which rows feed which gradient, accumulation order, strided views) and
(b) the dual-number treatment of the one-sided gradient penalty (torch double-backward is the ground truth),
independently of any GPU.  The same orchestration code runs on the MI355X with HipKernels (tests -m gpu).
"""
import pytest
import torch

import sgg_amd  # noqa: F401
from oracle import sgg_oracle as O
from oracle.kernels_ref import RefKernels
from sgg_amd.step import GanStep

DT = torch.float64


def make_states(V, S, scale_emb=25.0):
    gp = O.init_params("G", V, S, dtype=DT, perturb=0.1)
    dp = O.init_params("D", V, S, dtype=DT, perturb=0.1)
    dp["W"] = dp["W"] * scale_emb          # push the slopes above 1 so the one-sided penalty is active
    return gp, dp


def rel_err(a, b, floor=1e-6):
    """max |a-b| relative to max|b|, with an absolute floor so exactly-cancelling gradients (e.g. the critic's
    decoder bias: +1 from the fake rows, -1 from the real rows) compare as equal."""
    return float((a - b).abs().max() / (b.abs().max() + floor))


@pytest.fixture(scope="module", params=[32, 29])     # 29: odd maps (29, 15) on even canvases (trunk.plan_canvas), then 8, 4, 2
def setup(request):
    B, S, V = 2, request.param, 11
    gp, dp = make_states(V, S)
    images, labels, onehot = O.synth_batch(B, S, V, dtype=DT)
    K = RefKernels()
    gs = GanStep(K, V, S, B, lam=10.0, g_state=gp, d_state=dp, dtype=DT)
    return dict(B=B, S=S, V=V, gp=gp, dp=dp, images=images, labels=labels, onehot=onehot, gs=gs)


def test_forward_matches_oracle(setup):
    s = setup
    noise = O.synth_noise(s["B"], 0, DT)
    st, _ = s["gs"].generator_forward(s["images"], noise)
    ref = O.generator_forward(s["gp"], s["images"], noise)
    assert rel_err(st.OUT[0], ref) < 1e-10


def test_critic_and_generator_step_match_oracle(setup):
    s = setup
    B, gs = s["B"], s["gs"]
    gp = {k: v.clone() for k, v in s["gp"].items()}
    dp = {k: v.clone() for k, v in s["dp"].items()}
    noise0, noise1, alpha = O.synth_noise(B, 0, DT), O.synth_noise(B, 1, DT), O.synth_alpha(B, 0, DT)
    d_adam, g_adam = O.new_adam_state(dp), O.new_adam_state(gp)

    # ---- critic step ------------------------------------------------------------------------------------
    cost, aux, dgrads = O.d_step(gp, dp, d_adam, 1, s["images"], s["onehot"], noise0, alpha, lam=10.0)
    assert float(aux["gp"]) > 1e-3, "test needs an active gradient penalty (slopes %s)" % aux["slopes"]
    losses = gs.critic_step(s["images"], s["labels"], noise0, alpha.reshape(B))
    assert abs(float(losses[0]) - float(cost)) < 1e-9 * max(1.0, abs(float(cost)))
    assert abs(float(losses[2]) - float(aux["gp"])) < 1e-9 * max(1.0, float(aux["gp"]))
    for name, g in dgrads.items():
        assert g is not None, name
        e = rel_err(gs.D.grads[name], g)
        assert e < 1e-8, "critic grad %s rel err %.3e" % (name, e)
    for name in dp:
        if not O.is_dead(name):
            assert rel_err(gs.D.arena.views[name], dp[name]) < 1e-9, "critic param after Adam: " + name

    # ---- generator step (sees the updated critic) ---------------------------------------------------------
    gcost, gaux, ggrads = O.g_step(gp, dp, g_adam, 1, s["images"], noise1)
    glosses = gs.generator_step(s["images"], noise1)
    assert abs(-float(glosses[3]) - float(gcost)) < 1e-9 * max(1.0, abs(float(gcost)))
    for name, g in ggrads.items():
        e = rel_err(gs.G.grads[name], g)
        assert e < 1e-8, "generator grad %s rel err %.3e" % (name, e)
    for name in gp:
        if not O.is_dead(name):
            assert rel_err(gs.G.arena.views[name], gp[name]) < 1e-9, "generator param after Adam: " + name
    # dead branch untouched, argmax probe exact
    for name in gp:
        if O.is_dead(name):
            assert torch.equal(gs.G.arena.views[name], s["gp"][name])
    toks = gs.argmax_tokens(gs.G.head.state(1, B).OUT[0])
    assert torch.equal(toks, O.argmax_tokens(gaux["fake"]))

    # ---- a second critic step: Adam slots / step counter carry over ------------------------------------------
    noise2, alpha2 = O.synth_noise(B, 2, DT), O.synth_alpha(B, 1, DT)
    cost2, _, _ = O.d_step(gp, dp, d_adam, 2, s["images"], s["onehot"], noise2, alpha2, lam=10.0)
    losses2 = gs.critic_step(s["images"], s["labels"], noise2, alpha2.reshape(B))
    assert abs(float(losses2[0]) - float(cost2)) < 1e-8 * max(1.0, abs(float(cost2)))
    for name in dp:
        if not O.is_dead(name):
            assert rel_err(gs.D.arena.views[name], dp[name]) < 1e-8, "critic param after 2nd Adam: " + name


def test_generator_encoder_reuse_within_iteration_is_exact():
    """train.py's loop body (GanStep.iteration(reuse_g_encoder=True)): G's encoder once per iteration instead of CRITIC_ITERS + 1
    times.  Same minibatch, unchanged G weights: every weight after two iterations equals the recomputing schedule's bit for bit."""
    B, S, V, CI = 2, 32, 11, 2
    res = {}
    for reuse in (False, True):
        gp, dp = make_states(V, S)
        gs = GanStep(RefKernels(), V, S, B, lam=10.0, g_state=gp, d_state=dp, dtype=DT)
        calls = {"n": 0}
        fwd = gs.G.trunk.forward

        def counting(*a, _f=fwd, **k):
            calls["n"] += 1
            return _f(*a, **k)
        gs.G.trunk.forward = counting
        for it in range(2):
            images, labels, _ = O.synth_batch(B, S, V, seed_img=2 * it, seed_lab=2 * it + 1, dtype=DT)
            noises = [O.synth_noise(B, 10 * it + i, DT) for i in range(CI + 1)]
            alphas = [O.synth_alpha(B, 10 * it + i, DT).reshape(B) for i in range(CI)]
            gs.train_iteration(images, labels, noises, alphas, critic_iters=CI, reuse_g_encoder=reuse)
        gs.flush()
        assert calls["n"] == (2 if reuse else 2 * (CI + 1))
        assert gs._g_reuse is None and not gs._g_reuse_armed
        res[reuse] = [{k: v.clone() for k, v in net.arena.views.items()} for net in (gs.G, gs.D)]
    for a, b in zip(res[False], res[True]):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_critic_loss_without_update_matches_oracle_and_half_batch():
    """GanStep.critic_loss = the validation loss of train.py:375-377 (`sess.run(self.disc_cost, ...)` on a validation batch): the
    oracle's d_loss on the same inputs, no weight / gradient / Adam state touched; and the value on a VAL_BATCH_SIZE = B / 2 batch
    (train.py:30) is what the B-row machinery returns on that batch repeated twice (every term is a mean over rows)."""
    B, S, V = 4, 32, 11
    gp, dp = make_states(V, S)
    gs = GanStep(RefKernels(), V, S, B, lam=10.0, g_state=gp, d_state=dp, dtype=DT)
    images, labels, onehot = O.synth_batch(B, S, V, dtype=DT)
    noise, alpha = O.synth_noise(B, 0, DT), O.synth_alpha(B, 0, DT)
    w0 = [gs.G.arena.flat.clone(), gs.D.arena.flat.clone(), gs.D.grad_flat.clone(), gs.D.m_flat.clone()]
    got = gs.critic_loss(images, labels, noise, alpha.reshape(B)).clone()
    cost, aux = O.d_loss(gp, dp, images, onehot, noise, alpha, 10.0)
    assert float(aux["gp"]) > 1e-3
    assert abs(float(got[0]) - float(cost)) < 1e-9 and abs(float(got[2]) - float(aux["gp"])) < 1e-9
    assert abs(float(got[1]) - float(aux["wdist"])) < 1e-9
    for a, b in zip(w0, [gs.G.arena.flat, gs.D.arena.flat, gs.D.grad_flat, gs.D.m_flat]):
        assert torch.equal(a, b)
    assert gs.D.adam_t == 0 and gs.G.adam_t == 0
    # half batch repeated twice == the half batch's own loss
    h = B // 2
    rep = lambda t: torch.cat([t[:h], t[:h]])
    got2 = gs.critic_loss(rep(images), rep(labels), rep(noise), rep(alpha.reshape(B))).clone()
    cost_h, aux_h = O.d_loss(gp, dp, images[:h], onehot[:h], noise[:h], alpha[:h], 10.0)
    assert abs(float(got2[0]) - float(cost_h)) < 1e-9
    # a training step afterwards is unaffected by the evaluation in between
    gs.critic_step(images, labels, noise, alpha.reshape(B))
    gs2 = GanStep(RefKernels(), V, S, B, lam=10.0, g_state=gp, d_state=dp, dtype=DT)
    gs2.critic_step(images, labels, noise, alpha.reshape(B))
    assert torch.equal(gs.D.arena.flat, gs2.D.arena.flat)


def test_generator_encoder_reuse_rejects_a_modified_minibatch():
    B, S, V = 2, 32, 11
    gp, dp = make_states(V, S)
    gs = GanStep(RefKernels(), V, S, B, lam=10.0, g_state=gp, d_state=dp, dtype=DT)
    images, labels, _ = O.synth_batch(B, S, V, dtype=DT)
    with pytest.raises(AssertionError, match="modified in place"):
        with gs.iteration(reuse_g_encoder=True):
            gs.critic_step(images, labels, O.synth_noise(B, 0, DT), O.synth_alpha(B, 0, DT).reshape(B))
            images.mul_(1.5)                     # an augmentation / a loader refilling its buffer between two updates
            gs.critic_step(images, labels, O.synth_noise(B, 1, DT), O.synth_alpha(B, 1, DT).reshape(B))
