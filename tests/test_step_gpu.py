"""-m gpu: the full critic step + generator step on the MI355X (HIP kernels through the C ABI) against the CPU
oracle on identical seeded inputs (BASELINE.json configs[0]: 8 x 64x64 images, vocab 50).

Tolerances (fp32 on both sides, different summation orders):
  losses                          : |d| <= 2e-6 + 2e-6*|ref|      (tests/tolerances.py: ~10x what is measured)
  logits                          : |d| <= 1e-5 + 1e-5*max|ref|
  gradients                       : max|d| <= 2e-4 * max|ref| per tensor (+ floor 1e-7)
  arg-maxed triple tokens         : exact; the seeds' minimum top-2 logit margin exceeds 4x the logit tolerance
"""
import pytest
import torch

import sgg_amd  # noqa: F401
from oracle import sgg_oracle as O
from sgg_amd.step import GanStep
from tests.tolerances import GRAD_RTOL, MARGIN_FACTOR, logit_tol, loss_tol

pytestmark = pytest.mark.gpu


def tensor_err(a, b):
    return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-7))


def check_weights_after_adam(views, ref_params, ref_grads, old_params, t, what):
    """First-step Adam is sign-like: update = lr_t*g/(|g|*c + eps'), so an element whose gradient is at the fp32
    rounding-noise level may legitimately move by +-lr_t in either implementation.  Assert (a) every element
    moved by at most the Adam bound, (b) elements with a significant gradient (>= 1% of the tensor's max) got
    the oracle's update within 1% of lr_t."""
    lr_t = O.tf_adam_lr_t(t)
    bound = 1.05 * lr_t * (1 - O.ADAM_B1) / (1 - O.ADAM_B2) ** 0.5
    for n, g in ref_grads.items():
        w_hip, w_ref, w_old = views[n].cpu(), ref_params[n], old_params[n]
        assert float((w_hip - w_old).abs().max()) <= bound, "%s %s: update exceeds the Adam bound" % (what, n)
        sig = g.abs() >= 1e-2 * g.abs().max()
        if sig.any():
            d = ((w_hip - w_old) - (w_ref - w_old))[sig].abs().max()
            assert float(d) <= 1e-2 * lr_t, "%s %s: update differs by %.3e (lr_t %.3e)" % (what, n, float(d), lr_t)


@pytest.mark.parametrize("scale_emb", [1.0, 25.0])      # 25: slopes > 1, gradient penalty (second-order path) active
def test_gd_step_matches_oracle_config1(hip, scale_emb):
    B, S, V = 8, 64, 50
    gp, dp = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
    dp["W"] = dp["W"] * scale_emb
    images, labels, onehot = O.synth_batch(B, S, V)
    noise0, noise1, alpha = O.synth_noise(B, 0), O.synth_noise(B, 1), O.synth_alpha(B, 0)
    gs = GanStep(hip, V, S, B, lam=10.0, g_state=gp, d_state=dp)
    gp0, dp0 = {k: v.clone() for k, v in gp.items()}, {k: v.clone() for k, v in dp.items()}

    # forward parity of the generator alone
    st, _ = gs.generator_forward(images.cuda(), noise0.cuda())
    ref_logits = O.generator_forward(gp, images, noise0)
    assert float((st.OUT[0].cpu() - ref_logits).abs().max()) <= logit_tol(ref_logits.abs().max())

    d_adam, g_adam = O.new_adam_state(dp), O.new_adam_state(gp)
    cost, aux, dgrads = O.d_step(gp, dp, d_adam, 1, images, onehot, noise0, alpha)
    if scale_emb > 1:
        assert float(aux["gp"]) > 1e-3
    dl = gs.critic_step(images.cuda(), labels.cuda(), noise0.cuda(), alpha.reshape(B).cuda()).cpu()
    assert abs(float(dl[0]) - float(cost)) <= loss_tol(cost), (dl, cost)
    assert abs(float(dl[2]) - float(aux["gp"])) <= 1e-5 + 1e-4 * abs(float(aux["gp"])), (dl, aux["gp"])
    # the critic's decoder bias gradient cancels analytically (+1 from the fake rows, -1 from the real rows, 0 from GP)
    assert float(gs.D.grads["decoder/bias"].abs().max()) < 1e-5 and float(dgrads["decoder/bias"].abs().max()) < 1e-5
    worst = max((tensor_err(gs.D.grads[n], g), n) for n, g in dgrads.items() if n != "decoder/bias")
    assert worst[0] < GRAD_RTOL, "critic gradient %s: rel err %.3e" % (worst[1], worst[0])
    check_weights_after_adam(gs.D.arena.views, dp, {n: g for n, g in dgrads.items() if n != "decoder/bias"}, dp0, 1, "critic")

    # compare the generator step on IDENTICAL critic weights: the first Adam step moves noise-level gradient
    # elements by +-lr in either implementation (see check_weights_after_adam), which would otherwise leak
    # ~1e-4 of critic-output difference into this comparison.
    gs.D.arena.load_state_dict(dp)
    gs.D.trunk.refresh_weights()
    gcost, gaux, ggrads = O.g_step(gp, dp, g_adam, 1, images, noise1)
    gl = gs.generator_step(images.cuda(), noise1.cuda()).cpu()
    assert abs(-float(gl[3]) - float(gcost)) <= loss_tol(gcost)
    worst = max((tensor_err(gs.G.grads[n], g), n) for n, g in ggrads.items())
    assert worst[0] < GRAD_RTOL, "generator gradient %s: rel err %.3e" % (worst[1], worst[0])
    check_weights_after_adam(gs.G.arena.views, gp, ggrads, gp0, 1, "generator")

    toks = gs.argmax_tokens(gs.G.head.state(1, B).OUT[0]).cpu()
    margin = O.top2_margin(gaux["fake"])
    print("min top-2 logit margin: %.3e" % margin)
    assert margin > MARGIN_FACTOR * logit_tol(gaux["fake"].abs().max()), "seed gives an argmax margin inside the fp tolerance"
    assert torch.equal(toks, O.argmax_tokens(gaux["fake"]))



@pytest.mark.parametrize("mode,loss_tol,grad_tol", [(1, 3e-3, 3e-2), (4, 2e-2, 2e-1)], ids=["f16x1", "bf16x1"])
def test_single_piece_modes_step(hip, mode, loss_tol, grad_tol):
    """the mixed-precision conv modes (operands rounded to ONE fp16 / bf16 piece, one MFMA per product, f32
    accumulate; conv_precision 1 / 4).  Not the reference's arithmetic: a whole critic + generator step stays within the tolerance
    that 11 / 8 significant operand bits allow (losses relative, gradients relative to each tensor's maximum), the head
    (f32 MFMA GEMMs, VALU kernels) is unchanged."""
    B, S, V = 8, 64, 50
    old = hip.conv_precision
    hip.conv_precision = mode
    try:
        gp, dp = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
        dp["W"] = dp["W"] * 25.0
        images, labels, onehot = O.synth_batch(B, S, V)
        noise0, noise1, alpha = O.synth_noise(B, 0), O.synth_noise(B, 1), O.synth_alpha(B, 0)
        gs = GanStep(hip, V, S, B, lam=10.0, g_state=gp, d_state=dp)
        if getattr(hip, "conv_halo", True):       # (the default routing; SGG_OPTIONS=conv_halo=0 sends every layer to the gather kernels)
            assert gs.G.trunk.layers[1]["ws_layout"] == 1 and gs.G.trunk.layers[7]["ws_layout"] == 2      # the resident kernels serve it
        d_adam, g_adam = O.new_adam_state(dp), O.new_adam_state(gp)
        cost, aux, dgrads = O.d_step(gp, dp, d_adam, 1, images, onehot, noise0, alpha)
        dl = gs.critic_step(images.cuda(), labels.cuda(), noise0.cuda(), alpha.reshape(B).cuda()).cpu()
        assert abs(float(dl[0]) - float(cost)) <= loss_tol * max(1.0, abs(float(cost))), (dl, cost)
        worst = max((tensor_err(gs.D.grads[n], g), n) for n, g in dgrads.items() if n != "decoder/bias")
        print("mode %d critic gradients: worst rel err %.3e (%s)" % (mode, worst[0], worst[1]))
        assert worst[0] < grad_tol, "critic gradient %s: rel err %.3e" % (worst[1], worst[0])
        gs.D.arena.load_state_dict(dp)
        gs.D.trunk.refresh_weights()
        gcost, gaux, ggrads = O.g_step(gp, dp, g_adam, 1, images, noise1)
        gl = gs.generator_step(images.cuda(), noise1.cuda()).cpu()
        assert abs(-float(gl[3]) - float(gcost)) <= loss_tol * max(1.0, abs(float(gcost)))
        worst = max((tensor_err(gs.G.grads[n], g), n) for n, g in ggrads.items())
        print("mode %d generator gradients: worst rel err %.3e (%s)" % (mode, worst[0], worst[1]))
        assert worst[0] < grad_tol, "generator gradient %s: rel err %.3e" % (worst[1], worst[0])
    finally:
        hip.conv_precision = old


def test_ln_prologue_schedule_matches_unfused(hip):
    """LN prologue in the trunk: the LayerNorm + ELU of a layer is applied by the consumer's patch staging (forward, and - when a
    backward follows and the activation is never written - its wgrad).  K.ln_fusion = 2 fuses wherever the kernels allow (the default
    1 selects by a cost model that only pays at full size: tests/test_data_eval.py pins its decisions); same arithmetic up to the
    ELU's exp (|d| <= 1.2e-7): losses agree with the unfused schedule to 2e-5, every gradient to 6e-5."""
    if not getattr(hip, "conv_halo", True):
        pytest.skip("SGG_OPTIONS=conv_halo=0: no kernel with an LN prologue is in use")
    B, S, V = 8, 64, 50
    images, labels, _ = O.synth_batch(B, S, V)
    noise0, noise1, alpha = O.synth_noise(B, 0), O.synth_noise(B, 1), O.synth_alpha(B, 0)
    old = hip.ln_fusion
    res = {}
    try:
        for mode in (0, 2):
            hip.ln_fusion = mode
            gp, dp = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
            dp["W"] = dp["W"] * 25.0
            gs = GanStep(hip, V, S, B, lam=10.0, g_state=gp, d_state=dp)
            fused = [l["i"] for l in gs.G.trunk.layers if l.get("fuse_ln_bwd")]
            assert (len(fused) >= 4) == (mode == 2), fused
            dl = gs.critic_step(images.cuda(), labels.cuda(), noise0.cuda(), alpha.reshape(B).cuda()).clone()
            assert any(l.get("fused_now") for l in gs.G.trunk.layers) == (mode == 2)
            assert any(l.get("fused_now") for l in gs.D.trunk.layers) == (mode == 2)      # D's pass is followed by its backward
            gl = gs.generator_step(images.cuda(), noise1.cuda()).clone()
            gs.flush()
            res[mode] = (dl.cpu(), gl.cpu(), {k: v.clone().cpu() for k, v in gs.D.grads.items()}, {k: v.clone().cpu() for k, v in gs.G.grads.items()})
    finally:
        hip.ln_fusion = old
    a, b = res[0], res[2]
    for which in (0, 1):       # loss vectors (disc_cost, wasserstein, gp, mean D(fake))
        assert float((a[which] - b[which]).abs().max()) <= 1e-5 + 2e-5 * float(a[which].abs().max()), (a[which], b[which])
    for which in (2, 3):
        worst = max((float((a[which][k] - b[which][k]).abs().max() / (a[which][k].abs().max() + 1e-7)), k) for k in a[which])
        print("fused vs unfused gradients: worst rel diff %.3e (%s)" % worst)
        assert worst[0] < 6e-5, worst        # (2e-5 .. 3.3e-5 observed, by the rounding of the kernels in front; the parity bound is 1e-3)


@pytest.mark.gpu
def test_step_runs_the_kernels_the_routing_names(hip):
    """Which 3x3 kernels a whole G+D step launches for the 128-column layers (symbols as the library's own dispatch reports them to the
    timing hook, i.e. the native path that ran): the producer / consumer kernel for every dgrad and every forward without LN prologue,
    the four-wave kernel for forwards WITH the prologue (trunk._query_layouts), never the gather kernels on these shapes."""
    if not (getattr(hip, "halo_pc", True) and getattr(hip, "conv_halo", True) and getattr(hip, "presplit", True)):
        pytest.skip("the routing asserted here is the default one (SGG_OPTIONS switched the producer / consumer path off)")
    B, S, V = 8, 64, 50
    images, labels, _ = O.synth_batch(B, S, V)
    noise0, noise1, alpha = O.synth_noise(B, 0), O.synth_noise(B, 1), O.synth_alpha(B, 0)
    old = hip.ln_fusion
    try:
        hip.ln_fusion = 2       # (the default cost model fuses these layers at batch 64; 2 = wherever the kernels allow, also at batch 8)
        gs = GanStep(hip, V, S, B, lam=10.0, g_state=O.init_params("G", V, S, perturb=0.05), d_state=O.init_params("D", V, S, perturb=0.05))
        lay6 = gs.D.trunk.layers[6]
        assert (lay6["cin"], lay6["cout"], lay6["ws_layout"], lay6["ws_layout_bwd"]) == (128, 128, 1, 4), lay6["ws_layout"]
        hip.timing, hip.timing_conv_only, hip.timing_symbols = [], True, None
        gs.critic_step(images.cuda(), labels.cuda(), noise0.cuda(), alpha.reshape(B).cuda())
        gs.generator_step(images.cuda(), noise1.cuda())
        gs.flush()
        torch.cuda.synchronize()
        syms = [t[0] for t in hip.timing]
    finally:
        hip.timing, hip.ln_fusion = None, old
    count = lambda prefix: sum(s.startswith(prefix) for s in syms)
    # dgrads of conv2_4, conv3_1, conv3_2 in both networks (third template argument: the patch staged by LDS-DMA from a pre-split dy)
    assert count("conv_halo3_pc_kernel<true,false,") >= 6, sorted(set(syms))
    assert count("conv_halo3_kernel<2,128,2,2,true,true,false,true,false,1>") >= 6, sorted(set(syms))     # LN-prologue forwards
    assert count("conv_halo3_pc_kernel<true,true,") == 0 and count("conv_gather") == 0, sorted(set(syms))


@pytest.mark.gpu
@pytest.mark.parametrize("ln_mode", [0, 2])
def test_generator_encoder_reuse_within_iteration(hip, ln_mode):
    """train.py's loop (GanStep.iteration(reuse_g_encoder=True)): every update of an iteration sees the same minibatch and G's weights
    change only at its end, so G's encoder runs once per iteration instead of CRITIC_ITERS + 1 times.  Same kernels on the same
    inputs: with the LN prologue off the weights after two iterations are BIT-equal to the recomputing schedule; with it on, the
    critic updates read the with-backward fusion schedule instead of the forward-only one (ELU exp, |d| <= 1.2e-7)."""
    B, S, V, CI = 8, 64, 50, 3
    images, labels, _ = O.synth_batch(B, S, V)
    images2, labels2, _ = O.synth_batch(B, S, V, seed_img=2, seed_lab=3)
    old = hip.ln_fusion
    res = {}
    try:
        hip.ln_fusion = ln_mode
        for reuse in (False, True):
            gp, dp = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
            dp["W"] = dp["W"] * 25.0
            gs = GanStep(hip, V, S, B, lam=10.0, g_state=gp, d_state=dp)
            calls = {"n": 0}
            fwd = gs.G.trunk.forward

            def counting(*a, _f=fwd, **k):
                calls["n"] += 1
                return _f(*a, **k)
            gs.G.trunk.forward = counting
            for it, (im, lb) in enumerate(((images, labels), (images2, labels2))):
                im, lb = im.cuda().contiguous(), lb.cuda().contiguous()
                noises = [O.synth_noise(B, 10 * it + i).cuda() for i in range(CI + 1)]
                alphas = [O.synth_alpha(B, 10 * it + i).reshape(B).cuda() for i in range(CI)]
                gs.train_iteration(im, lb, noises, alphas, critic_iters=CI, reuse_g_encoder=reuse)
            gs.flush()
            assert calls["n"] == (2 if reuse else 2 * (CI + 1)), calls
            assert gs._g_reuse is None and not gs._g_reuse_armed
            res[reuse] = ({k: v.clone().cpu() for k, v in gs.G.arena.views.items()}, {k: v.clone().cpu() for k, v in gs.D.arena.views.items()})
    finally:
        hip.ln_fusion = old
    for net in (0, 1):
        for k in res[False][net]:
            a, b = res[False][net][k], res[True][net][k]
            if ln_mode == 0:
                assert torch.equal(a, b), (net, k, float((a - b).abs().max()))
            else:
                assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-6, (net, k)
