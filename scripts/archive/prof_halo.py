"""Cycle accounting of conv_halo3_kernel (instrumented build): python scripts/prof_halo.py B H Cin Cout [reps]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SGG_HIP_LIB", os.path.join(ROOT, "scene-graph-gan_amd", "_prof", "libsgg_hip_prof.so"))
sys.path.insert(0, ROOT)
import torch
import sgg_amd
from sgg_amd.lib import HipKernels

B, H, Ci, Co = [int(x) for x in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
K = HipKernels("cuda:0")
x = torch.randn((B, H, H, Ci), device="cuda"); w = torch.randn((3, 3, Ci, Co), device="cuda") * 0.05
b = torch.randn((Co,), device="cuda"); y = torch.empty((B, H, H, Co), device="cuda")
wf = torch.empty((3, 3, Co, Ci), device="cuda"); K.hwio_to_hwoi(w, wf)
am = torch.zeros(2, device="cuda"); K.absmax(x, am[0:1]); K.absmax(w, am[1:2])
ws = torch.empty((3, w.numel()), dtype=torch.int16, device="cuda"); K.split_weights(wf, ws, am[1:2], 1)
run = lambda: K.conv_fwd(x, w, wf, b, y, 1, ws, am[0:1], am[1:2], None, 1)
run(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
K.lib.sgg_halo_prof_read.restype = ctypes.c_int
K.lib.sgg_halo_prof_read(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
K.lib.sgg_halo_prof_read(out, 0)
v = list(out); n = max(v[6], 1)
ms = e0.elapsed_time(e1) / reps
print("B%d H%d %d->%d: %.3f ms/call %.1f TFLOP/s (instrumented)" % (B, H, Ci, Co, ms, 2.0 * B * H * H * Co * 9 * Ci / ms / 1e9))
names = ["total", "wait B (vmcnt)", "A reads (issue+lgkm)", "MFMA issue", "chunk boundary", "epilogue"]
for i, nm in enumerate(names): print("  %-22s %10.0f cycles/workgroup  %5.1f %%" % (nm, v[i] / n, 100.0 * v[i] / max(v[0], 1)))
print("  workgroup-launches", n)
