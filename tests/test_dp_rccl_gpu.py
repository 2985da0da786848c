"""-m gpu, needs TWO visible MI355X (skipped on the 1-GPU boxes of this pool): the data-parallel path over RCCL itself.

The reference is single-device (train.py:417-418); BASELINE.json configs[2] shards the minibatch over the GPUs of one node and
all-reduces G's and D's gradients with RCCL over xGMI.  tests/test_dp_gloo.py covers the orchestration on the CPU (gloo, fp64, the
kernel-level reference); these tests run the same thing on hardware the moment a box exposes two devices:
  (a) `python bench.py --gpus 2 --rendezvous-only` on the "nccl" (= RCCL) backend: the launch path the driver uses, one bucket-sized
      all-reduce through sgg_amd.dp.GradReducer;
  (b) two ranks with HipKernels, 4 rows each of one seeded 8-row draw, one critic + one generator update on the multi-stream
      schedule: replicas bit-equal, the all-reduced mean gradient == the single-process 8-row gradient (tests/tolerances.py), the
      weights after Adam within the Adam bound of the single-process ones.
Children are FRESH processes started before they touch a GPU (never an exec from this GPU-initialised process).
SGG_DP_REHEARSE_GLOO=1 runs both tests on a 1-GPU box with the two ranks sharing the card over gloo (what this build could execute:
profiles/r05_dp_rccl_test_gloo_rehearsal.log); without it they need two devices and the "nccl" backend."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REHEARSE = bool(os.environ.get("SGG_DP_REHEARSE_GLOO"))
BACKEND = "gloo" if REHEARSE else "nccl"
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2 and not REHEARSE, reason="needs two visible GPUs (RCCL over xGMI)")]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SGG_DP_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    if REHEARSE:
        env["SGG_DP_BACKEND"] = "gloo"
    return env


def test_bench_rendezvous_over_rccl():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True,
                         timeout=600, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["rccl"]["nranks"] == 2 and rec["rccl"]["backend"] == BACKEND and rec["rccl"]["allreduce_mean_ok"]


def test_two_rank_rccl_step_equals_single_process(hip, tmp_path):
    from oracle import sgg_oracle as O
    from sgg_amd.step import GanStep, tf_adam_lr_t
    from tests.tolerances import GRAD_RTOL, loss_tol
    prefix = str(tmp_path / "rank")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
           os.path.join(ROOT, "tests", "dp_rccl_worker.py"), prefix]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    r0, r1 = torch.load(prefix + "0.pt"), torch.load(prefix + "1.pt")
    assert r0["world"] == 2 and r0["backend"] == BACKEND
    for k in ("D.weights", "G.weights"):
        assert torch.equal(r0[k], r1[k]), "%s: the replicas diverged" % k
    for net in ("D", "G"):
        for n in r0[net + ".gradsum"]:
            assert torch.equal(r0[net + ".gradsum"][n], r1[net + ".gradsum"][n]), "%s gradient %s differs between the ranks after the all-reduce" % (net, n)
    # the single-process step on all 8 rows, same seeds (this process, cuda:0)
    Bg, S, V = 8, 64, 50
    gp, dp_ = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
    dp_["W"] = dp_["W"] * 25.0
    images, labels, _ = O.synth_batch(Bg, S, V)
    gs = GanStep(hip, V, S, Bg, lam=10.0, g_state=gp, d_state=dp_, overlap_streams=True)
    img = images.cuda()
    gs.critic_step(img, labels.cuda(), O.synth_noise(Bg, 0).cuda(), O.synth_alpha(Bg, 0).reshape(Bg).cuda())
    gs.generator_step(img, O.synth_noise(Bg, 1).cuda())
    gs.flush()
    torch.cuda.synchronize()
    for net, N in (("D", gs.D), ("G", gs.G)):
        worst = max((float((r0[net + ".gradsum"][n] * 0.5 - g.cpu()).abs().max() / (g.abs().max().cpu() + 1e-7)), n)
                    for n, g in N.grads.items() if not (net == "D" and n == "decoder/bias") and float(g.abs().max()) > 0)
        assert worst[0] <= GRAD_RTOL, "%s: mean of the per-rank gradients vs the 8-row gradient, %s: %.3e" % (net, worst[1], worst[0])
        # first Adam step is sign-like: an element may move by up to the Adam bound in either run (tests/test_step_gpu.py)
        bound = 1.05 * tf_adam_lr_t(1) * 0.5 / (0.1 ** 0.5)
        assert float((r0[net + ".weights"] - N.arena.flat.cpu()).abs().max()) <= 2 * bound
    # the losses are per-rank batch means: rank 0's critic loss is that of its own 4 rows - finite is all that can be said here
    assert bool(torch.isfinite(r0["losses"]).all()) and loss_tol(1.0) > 0
