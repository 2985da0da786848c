"""Per-shape timing of the head GEMMs (sgg_gemm_skinny_*) at configs[1]: python scripts/prof_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sgg_amd  # noqa: F401
from sgg_amd.lib import HipKernels

K = HipKernels("cuda:0")
# (name, mode, M, N, K, calls per G+D step)
SHAPES = [
    ("c.W_c fwd  R=64", "nn", 64, 196, 512, 9), ("c.W_c fwd  R=192", "nn", 192, 196, 512, 3), ("c.W_c fwd dual 128", "nn", 128, 196, 512, 3),
    ("gates G R=64", "nn", 64, 2048, 1536, 6), ("gates D R=192", "nn", 192, 2048, 1324, 3), ("gates D dual 128", "nn", 128, 2048, 1324, 3),
    ("gates D R=64", "nn", 64, 2048, 1324, 3),
    ("decoder G", "nn", 64, 1000, 512, 6), ("decoder D R=192", "nn", 192, 1, 512, 3), ("embed D M=64", "nn", 64, 300, 1000, 12),
    ("gates dgrad D 192", "nt", 192, 1324, 2048, 3), ("gates dgrad G 64", "nt", 64, 1536, 2048, 3), ("gates dgrad 128", "nt", 128, 1324, 2048, 3),
    ("gates wgrad D", "tn", 1324, 2048, 128, 3), ("gates wgrad G", "tn", 1536, 2048, 64, 3), ("decoder wgrad G", "tn", 512, 1000, 64, 3),
    ("W_c wgrad", "tn", 512, 196, 128, 9), ("dE.W_c^T", "nt", 192, 512, 196, 9), ("dec dgrad G", "nt", 64, 512, 1000, 3),
    ("emb dgrad (g)", "nt", 64, 1000, 300, 6), ("emb wgrad", "tn", 1000, 300, 64, 6),
]
# Direct ctypes calls with prebuilt arguments (the tensor-level wrappers cost ~ 11 us of host time per call, more than most of these
# kernels take): the loop below is GPU-bound for anything above ~ 3 us.
lib = K.lib
stream = torch.cuda.current_stream().cuda_stream
ws = K.workspace(256 << 20)
tot = 0.0
for name, mode, M, N, Kd, calls in SHAPES:
    C = torch.zeros((M, N), device="cuda")
    if mode == "nn":
        A, B = torch.randn((M, Kd), device="cuda"), torch.randn((Kd, N), device="cuda")
        args = (M, N, Kd, A.data_ptr(), Kd, B.data_ptr(), N, C.data_ptr(), N, None, 0, ws.data_ptr(), ws.numel(), stream)
        fn = lib.sgg_gemm_skinny_fwd
    elif mode == "nt":
        A, B = torch.randn((M, Kd), device="cuda"), torch.randn((N, Kd), device="cuda")
        args = (M, N, Kd, A.data_ptr(), Kd, B.data_ptr(), Kd, C.data_ptr(), N, 0, ws.data_ptr(), ws.numel(), stream)
        fn = lib.sgg_gemm_skinny_dgrad
    else:
        A, B = torch.randn((Kd, M), device="cuda"), torch.randn((Kd, N), device="cuda")
        args = (M, N, Kd, A.data_ptr(), M, B.data_ptr(), N, C.data_ptr(), N, 1, ws.data_ptr(), ws.numel(), stream)
        fn = lib.sgg_gemm_skinny_wgrad
    for _ in range(3):
        assert fn(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    tot += us * calls
    print("%-22s %s M%5d N%5d K%5d  %6.1f us  %6.1f TFLOP/s  x%d/step = %.0f us" % (name, mode, M, N, Kd, us, 2.0 * M * N * Kd / us / 1e6, calls, us * calls))
print("sum over listed calls: %.2f ms per step" % (tot / 1e3))
