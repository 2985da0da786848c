"""A faithful kernel timeline of the multi-stream G+D step under a profiler:   rocprofv3 --kernel-trace ... -- python3 scripts/trace_step.py [steps]

A profiler multiplies the host's cost per launch, so in a plain trace of bench.py the host falls behind in the head phases (chains
of 5 - 10 us launches) and streams that should start early start late (scripts/host_lead.py measures the untraced lead: the host
needs 8 ms to enqueue a 42 ms step).  Here every step is preceded by a GPU-side spin (torch.cuda._sleep, ~GATE_MS) on the main stream,
which every other stream's first operation of the step waits for: the host has enqueued the WHOLE step before the GPU starts it,
as it has in an untraced run.  The spin kernels delimit the steps in the trace (scripts/trace_timeline.py --gated)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sgg_amd.lib import HipKernels  # noqa: E402
from sgg_amd.params import init_state_dict  # noqa: E402
from sgg_amd.step import GanStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
GATE_MS = float(os.environ.get("GATE_MS", "80"))
B, S, V = bench.CONFIGS[1]
dev = torch.device("cuda:0")
K = HipKernels(dev)
K.conv_precision = 2
gs = GanStep(K, V, S, B, lam=10.0, g_state=init_state_dict("G", V, S), d_state=init_state_dict("D", V, S), overlap_streams=True)
images, labels, noises, alphas = bench.synth_inputs(B, S, V, 2 * (steps + 3), 0, 1, dev)


def one(k):
    gs.critic_step(images, labels, noises[2 * k], alphas[2 * k])
    gs.generator_step(images, noises[2 * k + 1])


for k in range(3):
    one(k)
gs.flush()
torch.cuda.synchronize(dev)
# cycles of the spin kernel per millisecond
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
torch.cuda._sleep(10_000_000)
e1.record()
torch.cuda.synchronize(dev)
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
gate = int(GATE_MS * cyc_per_ms)
for k in range(3, 3 + steps):
    torch.cuda._sleep(gate)
    t0 = time.perf_counter()
    one(k)
    host_ms = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize(dev)          # (the next spin starts on an idle GPU)
    print("step %d: host enqueue %.1f ms (gate %.0f ms)" % (k, host_ms, GATE_MS), flush=True)
gs.flush()
torch.cuda.synchronize(dev)
