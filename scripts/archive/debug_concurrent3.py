"""Main stream: a whole critic + generator step (every kernel of the path); side stream: a loop of one MFMA kernel on unrelated
buffers.  Every activation / gradient / weight is compared with the same step run alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa
from oracle import sgg_oracle as O
from sgg_amd.lib import HipKernels
from sgg_amd.step import GanStep

K = HipKernels("cuda:0")
B, S, V = 8, 64, 50
images, labels, _ = O.synth_batch(B, S, V)
img, lab = images.cuda(), labels.cuda()
noise0, noise1, alpha = O.synth_noise(B, 0).cuda(), O.synth_noise(B, 1).cuda(), O.synth_alpha(B, 0).reshape(B).cuda()
side = torch.cuda.Stream()

# aggressor operands: an independent network's buffers
agg = GanStep(K, V, S, B, lam=10.0, g_state=O.init_params("G", V, S, perturb=0.05), d_state=O.init_params("D", V, S, perturb=0.05))
agg.D.trunk.forward(img)
agg.critic_step(img, lab, noise0, alpha)
torch.cuda.synchronize()
AT = agg.D.trunk


def agg_conv(j, times):
    lay = AT.layers[j]
    x = img if j == 0 else AT.layers[j - 1]["a"]
    for _ in range(times):
        K.conv_fwd(x, lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"], lay["ws_fwd"], AT._am(0, j - 1) if j else None, AT._am(2, j), lay["tstats"],
                   lay["ws_layout"] if lay["ws_fwd"] is not None else 0)


def agg_dgrad(j, times):
    lay = AT.layers[j]
    dy = torch.ones(lay["out_shape"], device="cuda")
    dx = torch.empty(lay["in_shape"], device="cuda")
    for _ in range(times):
        K.conv_dgrad(dy, lay["w"], dx, lay["s"], lay["ws_bwd"], None, AT._am(2, j), lay["ws_layout_bwd"])


def agg_wgrad(j, times):
    lay = AT.layers[j]
    dy = torch.ones(lay["out_shape"], device="cuda")
    for _ in range(times):
        K.conv_wgrad(AT.layers[j - 1]["a"], dy, lay["gw"], lay["s"])


AGG = {"none": lambda: None, "s2 fwd (layer 7)": lambda: agg_conv(7, 300), "s2 dgrad (layer 7)": lambda: agg_dgrad(7, 300),
       "halo fwd (layer 6)": lambda: agg_conv(6, 300), "wgrad halo (layer 6)": lambda: agg_wgrad(6, 200), "wgrad tr (layer 10)": lambda: agg_wgrad(10, 200)}


def victim_step(beside):
    gp, dp = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
    dp["W"] = dp["W"] * 25.0
    gs = GanStep(K, V, S, B, lam=10.0, g_state=gp, d_state=dp)
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        beside()
    gs.critic_step(img, lab, noise0, alpha)
    gs.generator_step(img, noise1)
    gs.flush()
    busy = not side.query()
    torch.cuda.synchronize()
    out = {}
    for n, net in (("G", gs.G), ("D", gs.D)):
        for j, lay in enumerate(net.trunk.layers):
            out["%s.y%d" % (n, j)] = lay["y"].clone()
            if lay["has_ln"]:
                out["%s.a%d" % (n, j)] = lay["a"].clone()
        for k, v in net.grads.items():
            out["%s.grad.%s" % (n, k)] = v.clone()
        out[n + ".weights"] = net.arena.flat.clone()
    out["TRI"] = gs.TRI.clone()
    return out, busy


ref, _ = victim_step(lambda: None)
for name, fn in AGG.items():
    nbad, firsts, busy_all = 0, {}, True
    for rep in range(12):
        got, busy = victim_step(fn)
        busy_all &= busy
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        if bad:
            nbad += 1
            firsts[bad[0]] = firsts.get(bad[0], 0) + 1
    print("beside %-22s: %2d of 12 steps differ; first differing tensor: %s  (side still busy at the end: %s)" % (name, nbad, firsts, busy_all), flush=True)
