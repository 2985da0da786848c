"""Does a hipGraph shorten the boundary between dependent tiny kernels?  200 dependent sgg_fill launches, eager vs captured.
python scripts/ubench/graph_gap.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sgg_amd  # noqa: F401
from sgg_amd.lib import HipKernels

K = HipKernels("cuda:0")
x = torch.zeros(4096, device="cuda")
A, B, C = torch.randn((64, 512), device="cuda"), torch.randn((512, 196), device="cuda"), torch.zeros((64, 196), device="cuda")
K.gemm_nn(A, B, C)          # allocate the workspace outside any capture
N = 200


def chain_fill():
    for i in range(N):
        K.fill(x, float(i))


def chain_gemm():
    for i in range(N // 2):
        K.gemm_nn(A, B, C)          # split-K gemm + slab reduce: two dependent launches


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


big = torch.zeros(512 << 20, device="cuda")      # a 2 GiB fill: ~1 ms of GPU work the host can run ahead of


def queued(f):
    """GPU time of the chain when its launches were enqueued while the GPU was still busy (the situation inside a training step)."""
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        K.fill(big, 1.0)
    e0.record()
    f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for name, chain in (("fill", chain_fill), ("gemm+reduce", chain_gemm)):
    eager = timeit(chain)
    print("%-12s queued behind 4 ms of GPU work: %.1f us for %d launches = %.2f us each" % (name, queued(chain), N, queued(chain) / N))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()                     # warm-up on the capture stream (workspace allocation of that stream)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain()
    graph = timeit(g.replay)
    t0 = time.perf_counter(); chain(); host_eager = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); host_graph = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize()
    print("%-12s %d dependent launches: eager %.1f us (%.2f us each), graph replay %.1f us (%.2f us each); host time eager %.0f us, "
          "replay %.0f us" % (name, N, eager, eager / N, graph, graph / N, host_eager, host_graph))
