"""Micro-driver for profiling one conv shape: python scripts/prof_conv.py B H Cin Cout k stride [reps] [mode]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sgg_amd
from sgg_amd.lib import HipKernels, same_pads

B, H, Ci, Co, k, s = [int(x) for x in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
K = HipKernels("cuda:0")
x = torch.randn((B, H, H, Ci), device="cuda")
if os.environ.get("SGG_PROF_ZERO") == "1": x.zero_()
if os.environ.get("SGG_PROF_ZERO") == "2": x.fill_(1.0)
w = torch.randn((k, k, Ci, Co), device="cuda") * 0.05
b = torch.randn((Co,), device="cuda")
Ho = same_pads(H, k, s)[0]
y = torch.empty((B, Ho, Ho, Co), device="cuda")
dy = torch.randn((B, Ho, Ho, Co), device="cuda")
dx = torch.empty_like(x); dw = torch.empty_like(w)
wf = torch.empty((k, k, Co, Ci), device="cuda"); K.hwio_to_hwoi(w, wf)
lay = K.conv_wsplit_layout(k, s, H, H, Ci, Co)
am = torch.zeros(2, device="cuda"); K.absmax(x, am[0:1]); K.absmax(w, am[1:2])
ws = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda")
lay_b = K.conv_wsplit_layout(k, s, H, H, Co, Ci)
amdy = torch.zeros(1, device="cuda"); K.absmax(dy, amdy)
if mode == "fwd_ws": K.split_weights(wf, ws, am[1:2], lay)
if mode == "dgrad_ws": K.split_weights(w, ws, am[1:2], lay_b)
ws0 = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda")
if mode == "fwd_gather_ws": K.split_weights(wf, ws0, am[1:2], 0)
if mode == "dgrad_gather_ws": K.split_weights(w, ws0, am[1:2], 0)
ln = None
if mode.endswith("_ln"):        # LN prologue timing: identity statistics (mean 0, rstd 1), gamma 1, beta 0 -> the staging applies ELU(x)
    stats = torch.zeros((B, 2), device="cuda"); stats[:, 1] = 1.0
    ln = (stats, torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda"))
    K.split_weights(wf, ws, am[1:2], lay)
ts = None
if mode == "fwd_stats":          # forward with the LayerNorm partials in the epilogue (conv1_1 as the trunk calls it)
    ts = torch.empty((B, K.conv_tile_stats_count((B, Ho, Ho, Co), Ci, k, s, lay), 4), device="cuda")
def run():
    if mode == "fwd": K.conv_fwd(x, w, wf, b, y, s)
    elif mode == "fwd_stats": K.conv_fwd(x, w, wf if Ci != 3 else w, b, y, s, tile_stats=ts)
    elif mode == "fwd_ws": K.conv_fwd(x, w, wf, b, y, s, ws, am[0:1], am[1:2], None, lay)
    elif mode == "fwd_ws_ln": K.conv_fwd(x, w, wf, b, y, s, ws, am[0:1], am[1:2], None, lay, ln=ln)
    elif mode == "wgrad_ln": K.conv_wgrad(x, dy, dw, s, am[0:1], amdy, ln=ln)
    elif mode == "wgrad": K.conv_wgrad(x, dy, dw, s, am[0:1], amdy)
    elif mode == "dgrad": K.conv_dgrad(dy, w, dx, s)
    elif mode == "dgrad_ws": K.conv_dgrad(dy, w, dx, s, ws, amdy, am[1:2], lay_b)
    elif mode == "fwd_gather_ws":
        K.conv_fwd(x, w, wf, b, y, s, ws0, am[0:1], am[1:2], None, 0)
    elif mode == "dgrad_gather_ws":
        K.conv_dgrad(dy, w, dx, s, ws0, amdy, am[1:2], 0)
    else: K.conv_wgrad(x, dy, dw, s)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * Ho * Ho * Co * k * k * Ci
print("%s B%d H%d %d->%d k%d s%d: %.3f ms  %.1f TFLOP/s" % (mode, B, H, Ci, Co, k, s, ms, fl / ms / 1e9))
