"""-m gpu: the kernels at the sizes only BASELINE.json configs[3] (vocab 70 000) and configs[4] (448x448: L = 784) reach,
against fp64, and the end-to-end critic / generator step at those shapes against the CPU oracle and the committed goldens.

  * attention product ctx_flat @ W_ctx at K = 100 352 (224x224) and K = 401 920 (448x448): 128-way split-K
    (generator_with_attention.py:15), forward / dgrad / wgrad;
  * decoder [512, 70000] (generator_with_attention.py:88) and embedding [70000, 300] (discriminator_with_attention.py:87)
    forward / dgrad / wgrad, and the one-hot row gather / scatter-add against the dense product;
  * LayerNorm over 448*448*32 = 6.4 M elements per sample (generator_with_attention.py:30);
  * B = 2 G forward + critic step + generator step at 448x448, at V = 70 000 and at the configs[1] layer shapes (224x224,
    V = 1000) vs oracle/sgg_oracle.py (tolerances of tests/test_step_gpu.py) and vs tests/golden/*_b2_golden.npz.
"""
import os

import numpy as np
import pytest
import torch

import sgg_amd  # noqa: F401
from oracle import sgg_oracle as O
from sgg_amd.step import GanStep
from tests.tolerances import GRAD_RTOL, logit_tol, loss_tol

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def g(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=g(seed), dtype=torch.float32) * scale


def close(hip_t, ref_t, rtol=2e-5, atol=1e-6, what=""):
    h, r = hip_t.detach().cpu().double(), ref_t.double()
    assert h.shape == r.shape, (what, h.shape, r.shape)
    assert torch.isfinite(h).all(), what + ": non-finite values"
    tol = atol + rtol * float(r.abs().max())
    err = float((h - r).abs().max())
    print("%-40s max err %.3e (tol %.3e)" % (what, err, tol))
    assert err <= tol, "%s: max err %.3e > tol %.3e" % (what, err, tol)


@pytest.mark.parametrize("cfg", [(64, 196, 100352), (32, 784, 401920)], ids=["K100352_224px", "K401920_448px"])
def test_attention_product_full_k(hip, cfg):
    B, L, LC = cfg
    ctx = rnd((B, LC), 1)
    W = rnd((LC, L), 2, 1.0 / 300.0)
    bias = rnd((L,), 3)
    dP = rnd((B, L), 4)
    ctxd, Wd = ctx.cuda(), W.cuda()
    P = torch.full((B, L), float("nan"), device="cuda")
    hip.attn_ctx_fwd(ctxd, Wd, bias.cuda(), P)
    W64 = W.double()
    close(P, ctx.double() @ W64 + bias.double(), what="attn_ctx_gemm_fwd K=%d" % LC)
    dctx0 = rnd((B, LC), 5)
    dctx = dctx0.cuda()
    hip.attn_ctx_dgrad(dP.cuda(), Wd, dctx, accumulate=True)
    close(dctx, dctx0.double() + dP.double() @ W64.t(), what="attn_ctx_gemm_dgrad")
    del W64
    dW = torch.zeros((LC, L), device="cuda")
    hip.attn_ctx_wgrad(ctxd, dP.cuda(), dW, accumulate=True)
    close(dW, ctx.double().t() @ dP.double(), what="attn_ctx_gemm_wgrad")


def test_vocab70k_decoder_and_embedding(hip):
    V, E, R, H = 70000, 300, 192, 512
    h, Wdec, bdec = rnd((R, H), 10), rnd((H, V), 11, 0.02), rnd((V,), 12, 0.1)
    out = torch.full((R, 3, V), float("nan"), device="cuda")
    hd, Wd = h.cuda(), Wdec.cuda()
    hip.gemm_nn(hd, Wd, out[:, 1, :], bdec.cuda())                         # decoder logits land in a [R,3,V] slab
    close(out[:, 1, :], h.double() @ Wdec.double() + bdec.double(), what="decoder fwd [512,70000]")
    dout = rnd((R, V), 13)
    dh = torch.full((R, H), float("nan"), device="cuda")
    hip.gemm_nt(dout.cuda(), Wd, dh)                                       # K = 70 000
    close(dh, dout.double() @ Wdec.double().t(), what="decoder dgrad K=70000")
    dW = torch.zeros((H, V), device="cuda")
    hip.gemm_tn(hd, dout.cuda(), dW, accumulate=True)
    close(dW, h.double().t() @ dout.double(), what="decoder wgrad")
    del out, dW, Wd
    # embedding: logits rows (dense) and one-hot rows (gather)
    Wemb = (torch.rand((V, E), generator=g(14)) * 0.2 - 0.1)
    tri = rnd((R, 3, V), 15)
    XH = torch.full((R, 812 + 512), float("nan"), device="cuda")
    trid, Wed = tri.cuda(), Wemb.cuda()
    hip.gemm_nn(trid[:, 2, :], Wed, XH[:, 512:812])                        # strided A (ld = 3V), K = 70 000
    close(XH[:, 512:812], tri[:, 2, :].double() @ Wemb.double(), what="embedding fwd K=70000")
    dE = rnd((R, E), 16)
    dtri = torch.full((R, V), float("nan"), device="cuda")
    hip.gemm_nt(dE.cuda(), Wed, dtri)
    close(dtri, dE.double() @ Wemb.double().t(), what="embedding dgrad")
    dWe = torch.zeros((V, E), device="cuda")
    hip.gemm_tn(trid[:, 2, :], dE.cuda(), dWe, accumulate=True)
    close(dWe, tri[:, 2, :].double().t() @ dE.double(), what="embedding wgrad [70000,300]")


@pytest.mark.parametrize("V", [50, 70000])
def test_embed_gather_equals_dense_onehot_product(hip, ref, V):
    E, R = 300, 64
    labels = torch.randint(0, V, (R, 3), generator=g(20))
    labels[5, 1] = labels[9, 1] = labels[40, 1] = labels[3, 1]            # rows sharing a label (scatter-add order)
    W = torch.rand((V, E), generator=g(21)) * 0.2 - 0.1
    Wd, lab = W.cuda(), labels.cuda()
    XH = torch.full((R, 812 + 512), float("nan"), device="cuda")
    hip.embed_gather_fwd(lab[:, 1], Wd, XH[:, 512:812])
    assert torch.equal(XH[:, 512:812].cpu(), W[labels[:, 1]]), "gather must be exact"
    oh = torch.empty((R, 3, V), device="cuda")
    hip.onehot(lab, oh)
    dense = torch.empty((R, E), device="cuda")
    hip.gemm_nn(oh[:, 1, :], Wd, dense)
    close(dense, W[labels[:, 1]].double(), rtol=1e-6, atol=1e-7, what="dense one-hot product")
    dY = rnd((R, 812), 22)
    dW0 = rnd((V, E), 23, 0.01)
    dW = dW0.cuda()
    hip.embed_gather_bwd(lab[:, 1], dY.cuda()[:, 512:812], dW)
    exp = dW0.double().clone()
    ref.embed_gather_bwd(labels[:, 1], dY[:, 512:812].double(), exp)
    close(dW, exp, rtol=1e-6, atol=1e-7, what="embed_gather_bwd")
    # run twice from the same start: bit-identical (deterministic scatter-add)
    dW2 = dW0.cuda()
    hip.embed_gather_bwd(lab[:, 1], dY.cuda()[:, 512:812], dW2)
    assert torch.equal(dW, dW2)
    # label outside [0, V): zero row forward (tf.one_hot semantics), ignored backward
    bad = lab[:, 1].clone()
    bad[0] = V + 3
    hip.embed_gather_fwd(bad, Wd, XH[:, 512:812])
    assert float(XH[0, 512:812].abs().max()) == 0.0


def test_layernorm_448px_6M_elements_per_sample(hip, ref):
    shape = (2, 448, 448, 32)
    B, H, W, C = shape
    y = rnd(shape, 30, 2.0) + 0.3
    gamma, beta = 1.0 + rnd((C,), 31, 0.2), rnd((C,), 32, 0.2)
    da = rnd(shape, 33)
    a_ref = torch.empty(shape, dtype=torch.float64)
    st_ref = torch.empty((B, 2), dtype=torch.float64)
    ref.ln_elu_fwd(y.double(), gamma.double(), beta.double(), a_ref, st_ref)
    dy_ref = torch.empty(shape, dtype=torch.float64)
    dg_ref, db_ref, dbias_ref = (torch.empty(C, dtype=torch.float64) for _ in range(3))
    ref.ln_elu_bwd(y.double(), da.double(), gamma.double(), beta.double(), st_ref, dy_ref, dg_ref, db_ref, dbias_ref)
    yd, gd, bd, dad = y.cuda(), gamma.cuda(), beta.cuda(), da.cuda()
    a = torch.full(shape, float("nan"), device="cuda")
    st = torch.empty((B, 2), device="cuda")
    hip.ln_elu_fwd(yd, gd, bd, a, st)
    close(a, a_ref, what="LN fwd 448x448x32")
    close(st, st_ref, what="LN stats")
    dy = torch.full(shape, float("nan"), device="cuda")
    dg, db, dbias = (torch.full((C,), float("nan"), device="cuda") for _ in range(3))
    hip.ln_elu_bwd(yd, dad, gd, bd, st, dy, dg, db, dbias)
    close(dy, dy_ref, rtol=5e-5, what="LN bwd dy")
    close(dg, dg_ref, rtol=5e-5, what="LN dgamma (6.4 M terms per sample)")
    close(db, db_ref, rtol=5e-5, what="LN dbeta")


def tensor_err(a, b):
    return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-7))


E2E = [("configs1_shape_b2_golden.npz", 2, 224, 1000), ("configs4_shape_b2_golden.npz", 2, 448, 50),
       ("configs3_vocab_b2_golden.npz", 2, 64, 70000), ("realdata_221px_b2_golden.npz", 2, 221, 1000)]


@pytest.mark.parametrize("gold,B,S,V", E2E, ids=["configs1_224px_V1000", "configs4_448px_L784", "configs3_V70000", "realdata_221px_canvas"])
def test_step_matches_oracle_and_golden(hip, gold, B, S, V):
    """G forward + one critic update + one generator update on B = 2 rows at the layer / vocabulary sizes of configs[1], [4], [3]
    (default conv precision of the product path) vs the CPU oracle run here on the same seeded inputs, and vs the committed
    golden vectors the oracle produced (tests/golden/make_golden.py)."""
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    gp, dp = O.init_params("G", V, S), O.init_params("D", V, S)
    images, labels, onehot = O.synth_batch(B, S, V)
    noise0, noise1, alpha = O.synth_noise(B, 0), O.synth_noise(B, 1), O.synth_alpha(B, 0)
    gs = GanStep(hip, V, S, B, lam=10.0, g_state=gp, d_state=dp)
    if S == 221:     # odd maps run on even canvases (trunk.plan_canvas) through the same halo / band kernels as 224x224
        assert gs.G.trunk.img_canvas is not None and gs.G.trunk.layers[0]["in_shape"][1] == 224
        if getattr(hip, "conv_halo", True):      # (the default routing)
            assert gs.G.trunk.layers[1]["ws_layout"] == 1 and gs.G.trunk.layers[7]["ws_layout"] == 2
    G = np.load(os.path.join(GOLD_DIR, gold))
    st, _ = gs.generator_forward(images.cuda(), noise0.cuda())
    logits = st.OUT[0].cpu()
    ref_logits = O.generator_forward(gp, images, noise0)
    tol = logit_tol(ref_logits.abs().max())
    assert float((logits - ref_logits).abs().max()) <= tol
    gl = G["g_logits_step0"]
    sub = logits.numpy() if gl.shape == tuple(logits.shape) else logits.numpy()[..., ::97]
    assert np.abs(sub - gl).max() <= tol, "generator logits vs golden"
    toks0 = gs.argmax_tokens(st.OUT[0]).cpu()
    assert torch.equal(toks0, O.argmax_tokens(ref_logits)) and np.array_equal(toks0.numpy(), G["g_tokens_step0"])

    d_adam, g_adam = O.new_adam_state(dp), O.new_adam_state(gp)
    cost, aux, dgrads = O.d_step(gp, dp, d_adam, 1, images, onehot, noise0, alpha)
    dl = gs.critic_step(images.cuda(), labels.cuda(), noise0.cuda(), alpha.reshape(B).cuda()).cpu()
    assert abs(float(dl[0]) - float(cost)) <= loss_tol(cost), (dl, cost)
    assert abs(float(dl[0]) - float(G["disc_cost"])) <= 10 * loss_tol(G["disc_cost"])     # (golden: the oracle on another host)
    assert abs(float(dl[2]) - float(aux["gp"])) <= 1e-5 + 1e-4 * abs(float(aux["gp"]))
    worst = max((tensor_err(gs.D.grads[n], gr), n) for n, gr in dgrads.items() if n != "decoder/bias")
    print("critic gradients: worst rel err %.3e (%s)" % worst)
    assert worst[0] < GRAD_RTOL, "critic gradient %s: rel err %.3e" % (worst[1], worst[0])

    gs.D.arena.load_state_dict(dp)            # generator step on identical critic weights (tests/test_step_gpu.py)
    gs.D.trunk.refresh_weights()
    gcost, gaux, ggrads = O.g_step(gp, dp, g_adam, 1, images, noise1)
    glv = gs.generator_step(images.cuda(), noise1.cuda()).cpu()
    assert abs(-float(glv[3]) - float(gcost)) <= loss_tol(gcost)
    assert abs(-float(glv[3]) - float(G["gen_cost"])) <= 10 * loss_tol(G["gen_cost"])
    worst = max((tensor_err(gs.G.grads[n], gr), n) for n, gr in ggrads.items())
    print("generator gradients: worst rel err %.3e (%s)" % worst)
    assert worst[0] < GRAD_RTOL, "generator gradient %s: rel err %.3e" % (worst[1], worst[0])
    toks = gs.argmax_tokens(gs.G.head.state(1, B).OUT[0]).cpu()
    margin = O.top2_margin(gaux["fake"])
    print("min top-2 logit margin: %.3e" % margin)
    assert torch.equal(toks, O.argmax_tokens(gaux["fake"])) and np.array_equal(toks.numpy(), G["g_tokens_step1"])
