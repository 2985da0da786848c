"""CPU: the data-parallel overlap order of step.py, pinned with a recording reducer (no process group needed).

Every gradient all-reduce (launched by Network.update at the end of a critic / generator step) must have the OTHER
network's encoder forward enqueued before anything waits for it (Network.finish_update -> PendingReduce.wait): that
forward is the work the collective hides under (Multi-GPU; train.py:362-368 is the loop whose two updates
the overlap sits between)."""
import pytest
import torch

import sgg_amd  # noqa: F401
from oracle import sgg_oracle as O
from oracle.kernels_ref import RefKernels
from sgg_amd.step import GanStep

DT = torch.float64


class _Pending:
    def __init__(self, log, kind):
        self.log, self.kind = log, kind

    def wait(self):
        self.log.append("wait:" + self.kind)
        return 1.0


def _recording_step(critic_iters, iterations=2):
    B, S, V = 2, 32, 7
    log = []

    def reducer(net):
        log.append("reduce:" + net.kind)
        return _Pending(log, net.kind)

    gs = GanStep(RefKernels(), V, S, B, g_state=O.init_params("G", V, S, dtype=DT), d_state=O.init_params("D", V, S, dtype=DT),
                 dtype=DT, reducer=reducer)
    for net in (gs.G, gs.D):
        fwd = net.trunk.forward

        def wrapped(images, *args, _f=fwd, _k=net.kind):
            log.append("encoder:" + _k)
            return _f(images, *args)

        net.trunk.forward = wrapped
    images, labels, _ = O.synth_batch(B, S, V, dtype=DT)
    k = 0
    for _ in range(iterations):
        noises = [O.synth_noise(B, k + i, DT) for i in range(critic_iters + 1)]
        alphas = [O.synth_alpha(B, k + i, DT).reshape(B) for i in range(critic_iters)]
        gs.train_iteration(images, labels, noises, alphas, critic_iters)
        k += critic_iters + 1
    gs.flush()
    return log, gs


@pytest.mark.parametrize("critic_iters", [1, 3])
def test_every_reduce_is_covered_by_the_other_encoder(critic_iters):
    log, gs = _recording_step(critic_iters)
    other = {"G": "D", "D": "G"}
    n_reduce = 0
    for i, ev in enumerate(log):
        if not ev.startswith("reduce:"):
            continue
        n_reduce += 1
        kind = ev.split(":")[1]
        j = log.index("wait:" + kind, i)                     # the wait that consumes this reduce
        between = log[i + 1:j]
        assert "reduce:" + kind not in between, "a second reduce of %s was launched before the first was waited for" % kind
        if j == len(log) - 1 or all(e.startswith("wait:") for e in log[j:]):
            continue                                         # consumed by the final flush(): nothing left to overlap with
        assert "encoder:" + other[kind] in between, (
            "the %s-gradient all-reduce is waited for before %s's encoder forward was enqueued: %s" % (kind, other[kind], log[i:j + 1]))
    assert n_reduce == 2 * (critic_iters + 1)
    assert gs.D.adam_t == 2 * critic_iters and gs.G.adam_t == 2
