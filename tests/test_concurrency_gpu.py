"""-m gpu: kernels of the path keep their results when another HIP stream runs MFMA kernels beside them.

Found in round 2: with packed-fp32 VALU instructions (v_pk_fma_f32) in the library, conv_c3_fwd_kernel and
conv_c3_wgrad_kernel returned wrong low halves in lanes 48-63 whenever conv_s2_kernel (or, rarely, conv_halo3_kernel) of ANOTHER
stream shared their SIMD - 116 of 120 outputs differed; the single-stream schedule never shows it.  The library is built
without packed-fp32 instructions (sgg_amd/build.py); this test keeps every kernel of a critic + generator step honest beside the
heaviest MFMA kernels, and the two-stream schedule (GanStep(overlap_streams=True)) bit-equal to the serial one."""
import pytest
import torch

import sgg_amd  # noqa: F401
from oracle import sgg_oracle as O
from sgg_amd.step import GanStep

pytestmark = pytest.mark.gpu
B, S, V = 8, 64, 50


def _inputs():
    images, labels, _ = O.synth_batch(B, S, V)
    return (images.cuda(), labels.cuda(), O.synth_noise(B, 0).cuda(), O.synth_noise(B, 1).cuda(), O.synth_alpha(B, 0).reshape(B).cuda())


def _new_step(hip, **kw):
    gp, dp = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
    dp["W"] = dp["W"] * 25.0
    return GanStep(hip, V, S, B, lam=10.0, g_state=gp, d_state=dp, **kw)


def _snapshot(gs):
    torch.cuda.synchronize()
    out = {}
    for n, net in (("G", gs.G), ("D", gs.D)):
        for j, lay in enumerate(net.trunk.layers):
            out["%s.y%d" % (n, j)] = lay["y"].clone()
        for k, v in net.grads.items():
            out["%s.grad.%s" % (n, k)] = v.clone()
        out[n + ".weights"] = net.arena.flat.clone()
    out["losses"] = torch.cat([gs.d_losses, gs.g_losses]).clone()
    return out


def test_step_is_unchanged_beside_mfma_kernels_of_another_stream(hip):
    if not getattr(hip, "conv_halo", True):
        pytest.skip("SGG_OPTIONS=conv_halo=0: the aggressor of this test is the resident MFMA kernels' staging variant")
    img, lab, noise0, noise1, alpha = _inputs()
    side = torch.cuda.Stream()
    agg = _new_step(hip)                       # an independent network: operands of the kernels that run beside
    agg.critic_step(img, lab, noise0, alpha)
    torch.cuda.synchronize()
    T = agg.D.trunk

    def conv(j, times):
        lay = T.layers[j]
        for _ in range(times):
            # (the input as the step's own forward hands it over: pre-split where the LayerNorm kernel wrote it so - the aggressor is
            #  then the LDS-DMA staging variant the default schedule runs, on valid operands)
            hip.conv_fwd(T.layers[j - 1]["a"], lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"], lay["ws_fwd"], T._am(0, j - 1), T._am(2, j),
                         lay["tstats"], lay["ws_layout"], **({"x_s16": True} if T.layers[j - 1].get("a_s16_now") else {}))

    def dgrad(j, times):
        lay = T.layers[j]
        dy, dx = torch.ones(lay["out_shape"], device="cuda"), torch.empty(lay["in_shape"], device="cuda")
        for _ in range(times):
            hip.conv_dgrad(dy, lay["w"], dx, lay["s"], lay["ws_bwd"], None, T._am(2, j), lay["ws_layout_bwd"])

    assert T.layers[7]["ws_layout"] == 2 and T.layers[6]["ws_layout"] in (1, 4)     # (4: the producer / consumer 3x3 kernel)

    def run(beside):
        gs = _new_step(hip)
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            beside()
        gs.critic_step(img, lab, noise0, alpha)
        gs.generator_step(img, noise1)
        gs.flush()
        overlapped = not side.query()          # the other stream was still busy when the step had been enqueued
        return _snapshot(gs), overlapped

    ref, _ = run(lambda: None)
    for name, beside in (("conv_s2 forward", lambda: conv(7, 300)), ("conv_s2 dgrad", lambda: dgrad(7, 300)), ("conv_halo3 forward", lambda: conv(6, 300))):
        for rep in range(4):
            got, overlapped = run(beside)
            assert overlapped, "the kernels beside finished before the step was enqueued: nothing was tested"
            bad = [k for k in ref if not torch.equal(ref[k], got[k])]
            assert not bad, "beside %s (repetition %d) %d tensors differ, first %s" % (name, rep, len(bad), bad[:3])


@pytest.mark.parametrize("ln_fusion,size", [(1, (8, 64, 50)), (2, (8, 64, 50)), (1, (64, 224, 1000))],
                         ids=["default", "ln_prologue_everywhere", "configs1_full_size"])
def test_two_stream_schedule_is_bitwise_the_serial_one(hip, ln_fusion, size):
    """overlap_streams=True (D's encoder beside G's forward; filter gradients beside the dgrad -> LayerNorm-backward chain; the heads'
    parameter-gradient work deferred to a third stream; G's forward of the next update early on a fourth) only reorders independent
    launches: after two iterations of two critic updates each (train.py:362-368), every weight equals the serial run's.  With
    K.ln_fusion = 2 the filter gradients on the side stream also apply the LayerNorm prologue (the activation was never written).
    configs1_full_size: the same at BASELINE.json configs[1] (batch 64, 224x224, vocab 1000) - the size bench.py times, where a 0.8 ms
    conv_s2 launch overlaps a 0.2 ms LayerNorm pass instead of microsecond kernels (bench.py records the same check in its line as
    parity.two_stream_bitwise_at_full_size)."""
    b, s_, v = size
    images, labels, _ = O.synth_batch(b, s_, v)
    img, lab = images.cuda(), labels.cuda()
    old_fusion = hip.ln_fusion
    hip.ln_fusion = ln_fusion

    def run(overlap):
        gp, dp = O.init_params("G", v, s_, perturb=0.05), O.init_params("D", v, s_, perturb=0.05)
        dp["W"] = dp["W"] * 25.0
        gs = GanStep(hip, v, s_, b, lam=10.0, g_state=gp, d_state=dp, overlap_streams=overlap)
        for it in range(2):
            noises = [O.synth_noise(b, 10 * it + i).cuda() for i in range(3)]
            alphas = [O.synth_alpha(b, 10 * it + i).reshape(b).cuda() for i in range(2)]
            gs.train_iteration(img, lab, noises, alphas, critic_iters=2)
        gs.flush()
        snap = _snapshot(gs)
        early = getattr(gs, "xs", None) is not None
        del gs
        torch.cuda.empty_cache()
        return snap, early

    try:
        ref, _ = run(False)
        assert all(bool(torch.isfinite(t).all()) for t in ref.values())
        for rep in range(3 if b == 8 else 1):
            got, early = run(True)
            assert early == bool(getattr(hip, "g_early", 0)), "the early stream of G's forward did not take part"
            bad = [k for k in ref if not torch.equal(ref[k], got[k])]
            assert not bad, "repetition %d: %d tensors differ, first %s" % (rep, len(bad), bad[:3])
    finally:
        hip.ln_fusion = old_fusion


def test_step_is_unchanged_beside_collective_like_traffic(hip):
    """Data parallel runs RCCL's all-reduce kernels on their own stream beside EVERY kernel of the step (sgg_amd/dp.py: the critic's
    reduce under G's forward, the generator's under D's encoder).  A ring all-reduce is, per hop, a copy of a bucket chunk and a
    reduction (add) into the bucket, repeated; RCCL itself needs a second GPU, so its stand-ins here are large `copy_` / `add_`
    launches over 64 MB buckets (the GradReducer bucket size) looping on a second stream for the whole critic + generator step:
    HBM-saturating vector kernels beside the MFMA kernels - the other side of the packed-fp32 finding above.  Every output, every
    gradient and every weight must equal the serial step's bit for bit."""
    img, lab, noise0, noise1, alpha = _inputs()
    side = torch.cuda.Stream()
    n = (64 << 20) // 4
    bucket, recv = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    bucket2 = torch.zeros(n, device="cuda")

    def ring(hops):
        # (whole 64 MB buckets per launch: ~25 us of HBM-saturating traffic each, so the GPU side outlasts the host's enqueue of the step)
        for h in range(hops):
            recv.copy_(bucket)                 # "receive" the neighbour's bucket
            bucket2.add_(recv)                 # reduce it into the local one
            bucket.copy_(bucket2)              # forward the partial sum
        bucket2.mul_(1.0 / 8.0)

    def run(beside):
        gs = _new_step(hip)
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            beside()
        gs.critic_step(img, lab, noise0, alpha)
        gs.generator_step(img, noise1)
        gs.flush()
        overlapped = not side.query()
        return _snapshot(gs), overlapped

    ref, _ = run(lambda: None)
    for rep in range(4):
        got, overlapped = run(lambda: ring(1500))
        assert overlapped, "the collective-like traffic finished before the step was enqueued: nothing was tested"
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, "beside ring-all-reduce-like traffic (repetition %d) %d tensors differ, first %s" % (rep, len(bad), bad[:3])
    # and on the two-stream schedule (three streams busy)
    gs_ref = _new_step(hip, overlap_streams=True)
    gs_ref.critic_step(img, lab, noise0, alpha)
    gs_ref.generator_step(img, noise1)
    gs_ref.flush()
    ref2 = _snapshot(gs_ref)
    gs = _new_step(hip, overlap_streams=True)
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ring(1500)
    gs.critic_step(img, lab, noise0, alpha)
    gs.generator_step(img, noise1)
    gs.flush()
    assert not side.query()
    got2 = _snapshot(gs)
    bad = [k for k in ref2 if not torch.equal(ref2[k], got2[k])]
    assert not bad, "two-stream schedule beside ring-like traffic: %d tensors differ, first %s" % (len(bad), bad[:3])
    assert all(torch.equal(ref[k], ref2[k]) for k in ref)


def test_head_side_stream_is_bitwise_the_single_stream_schedule(hip):
    """head.py runs everything off the recurrent dependency chain (embedding / decoder products, every parameter-gradient GEMM and
    column sum of the heads, the attention weights' gradient) on a second stream beside the chain.  Same kernels, same operands,
    same accumulation order: after two iterations of two critic updates every weight, gradient and activation equals the
    single-stream schedule's, also in combination with the two-stream encoder schedule."""
    img, lab = _inputs()[:2]

    def run(**kw):
        gs = _new_step(hip, **kw)
        for it in range(2):
            noises = [O.synth_noise(B, 10 * it + i).cuda() for i in range(3)]
            alphas = [O.synth_alpha(B, 10 * it + i).reshape(B).cuda() for i in range(2)]
            gs.train_iteration(img, lab, noises, alphas, critic_iters=2)
        gs.flush()
        return _snapshot(gs)

    ref = run(head_side_stream=False)
    for kw in ({"head_side_stream": True}, {"head_side_stream": True, "overlap_streams": True}):
        for rep in range(3):
            got = run(**kw)
            bad = [k for k in ref if not torch.equal(ref[k], got[k])]
            assert not bad, "%s repetition %d: %d tensors differ, first %s" % (kw, rep, len(bad), bad[:3])
