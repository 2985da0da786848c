"""Debug aid: run training iterations with every ln_elu_bwd followed by a look at the one-pass kernel's counter lines
(bash scripts/build_variant_one.sh lnbdbg layernorm.hip -DSGG_LN_BWD_FUSED=1 -DSGG_LNB_DEBUG; SGG_HIP_LIB=.../_prof/libsgg_hip_lnbdbg.so).
Error line words: flag, workgroups that gave up, then - from the first of them - sample, arrivals seen, own ticket, G, arrivals and
tickets taken at that moment."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa: E402,F401
from oracle import sgg_oracle as O  # noqa: E402
from sgg_amd.lib import HipKernels  # noqa: E402
from sgg_amd.step import GanStep  # noqa: E402

LINE = 32


def counters(ws, B, HW, C):
    N = HW * C
    nch = -(-N // 4096)
    G0 = min(-(-1536 // B), nch)
    cpg = -(-nch // G0)
    G0 = -(-nch // cpg)
    Gf = -(-N // 16384)
    off = 4 * (B * G0 * 4 + B * Gf * 2 + B * Gf * 3 * C)
    base = ws.data_ptr()
    off = ((base + off + 127) & ~127) - base
    n = (2 * B + 2) * LINE
    return ws[off:off + 4 * n].view(torch.int32).cpu().view(2 * B + 2, LINE), Gf


def main():
    K = HipKernels("cuda:0")
    B, S, V = 64, 224, 7004
    images, labels, _ = O.synth_batch(B, S, V)
    img, lab = images.cuda(), labels.cuda()
    gs = GanStep(K, V, S, B, lam=10.0, g_state=O.init_params("G", V, S), d_state=O.init_params("D", V, S))
    orig = K.ln_elu_bwd
    state = {"n": 0, "bad": 0}

    def wrapped(y, da, gamma, beta, stats, dy, dgamma, dbeta, dbias_prev, amax_out=None, region=None, ws=None):
        orig(y, da, gamma, beta, stats, dy, dgamma, dbeta, dbias_prev, amax_out, region, ws)
        torch.cuda.synchronize()
        Bq, H, W, C = y.shape
        w = ws if ws is not None else K.workspace(0)
        c, G = counters(w, Bq, H * W, C)
        state["n"] += 1
        ok = int(c[0, 0]) >= Bq * G and bool((c[2:2 + Bq, 0] == G).all()) and int(c[1, 0]) == 0
        if not ok:
            state["bad"] += 1
            print("call %d shape %s G %d: ticket %d (want %d) err line %s" % (state["n"], tuple(y.shape), G, int(c[0, 0]), Bq * G, c[1, :9].tolist()), flush=True)
            print("   arrivals ", c[2:2 + Bq, 0].tolist(), flush=True)

    K.ln_elu_bwd = wrapped
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        noise = [O.synth_noise(B, 10 * it + k).cuda() for k in range(2)]
        alpha = O.synth_alpha(B, it).reshape(B).cuda()
        gs.critic_step(img, lab, noise[0], alpha)
        gs.generator_step(img, noise[1])
        print("iteration %d: %d calls, %d bad" % (it, state["n"], state["bad"]), flush=True)


if __name__ == "__main__":
    main()
