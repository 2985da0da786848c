"""Is the host ahead of the GPU in the multi-stream G+D step?   python scripts/host_lead.py [steps]

Without a profiler attached: (1) host time to ENQUEUE a step against the GPU time to run it; (2) how long after the event that
permits it (the critic update's G head has run) G's early encoder forward actually starts on its stream - a start that is late by
milliseconds means the host had not yet enqueued it (the head phases are then host-bound), a start within microseconds means the
host keeps ahead; (3) per phase of a critic update, the host's lead: the time between the host finishing the enqueue of the phase and
the GPU finishing its execution (negative = the GPU waited for the host)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sgg_amd.lib import HipKernels  # noqa: E402
from sgg_amd.step import GanStep  # noqa: E402
from sgg_amd.params import init_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, S, V = bench.CONFIGS[1]
dev = torch.device("cuda:0")
K = HipKernels(dev)
K.conv_precision = 2
gs = GanStep(K, V, S, B, lam=10.0, g_state=init_state_dict("G", V, S), d_state=init_state_dict("D", V, S), overlap_streams=True)
images, labels, noises, alphas = bench.synth_inputs(B, S, V, 2 * (steps + 3), 0, 1, dev)

# (2): time stamps around the early forward's start
marks = []
orig = gs._g_early_stream


def probed(images_):
    ev = getattr(gs, "_ev_g_free", None)
    xs = orig(images_)
    if xs is not None and ev is not None and getattr(ev, "_timed", False):
        e = torch.cuda.Event(enable_timing=True)
        e.record(xs)
        marks.append((ev, e))
    return xs


gs._g_early_stream = probed
_Event = torch.cuda.Event


class TimedEvent(_Event):          # step.py creates its events without timing: give them timing for this probe
    def __new__(cls, *a, **kw):
        kw["enable_timing"] = True
        o = super().__new__(cls, **kw)
        o._timed = True
        return o


torch.cuda.Event = TimedEvent


def one(k):
    gs.critic_step(images, labels, noises[2 * k], alphas[2 * k])
    gs.generator_step(images, noises[2 * k + 1])


for k in range(3):
    one(k)
gs.flush()
torch.cuda.synchronize(dev)
marks.clear()
t0 = time.perf_counter()
host = []
for k in range(3, 3 + steps):
    h0 = time.perf_counter()
    one(k)
    host.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
gs.flush()
torch.cuda.synchronize(dev)
t_all = time.perf_counter() - t0
print("host enqueue %.2f ms per step (min %.2f, max %.2f); GPU %.2f ms per step" %
      (1e3 * t_enq / steps, 1e3 * min(host), 1e3 * max(host), 1e3 * t_all / steps))
late = [ev.elapsed_time(e) for ev, e in marks]
print("G's early forward starts after its event by (ms):", " ".join("%.3f" % x for x in late))
