"""CPU, world_size 2 over gloo: the data-parallel path (sgg_amd/dp.py + the deferred-Adam overlap hooks of step.py).
Two ranks with B rows each of one global draw must reproduce the single-process 2B-row step: mean of per-rank
gradients == global-batch gradient, identical weights after Adam on both ranks.  fp64 with the kernel-level
reference injected (the GPU suite covers the same orchestration with HipKernels on one device)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sgg_amd  # noqa: F401
    from oracle import sgg_oracle as O
    from oracle.kernels_ref import RefKernels
    from sgg_amd import dp
    from sgg_amd.step import GanStep
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo")
    DT = torch.float64
    Bg, S, V = 4, 32, 11
    gp, dp_ = O.init_params("G", V, S, dtype=DT, perturb=0.1), O.init_params("D", V, S, dtype=DT, perturb=0.1)
    dp_["W"] = dp_["W"] * 25.0
    images, labels, _ = O.synth_batch(Bg, S, V, dtype=DT)
    noise0, noise1, alpha = O.synth_noise(Bg, 0, DT), O.synth_noise(Bg, 1, DT), O.synth_alpha(Bg, 0, DT).reshape(Bg)
    sh = lambda t: dp.shard_rows(t, rank, world)
    gs = GanStep(RefKernels(), V, S, Bg // world, g_state=gp, d_state=dp_, dtype=DT, reducer=dp.GradReducer())
    gs.critic_step(sh(images), sh(labels), sh(noise0), sh(alpha))
    gs.generator_step(sh(images), sh(noise1))
    # a second iteration with two critic updates: every deferred Adam step / reordered encoder of step.py is exercised
    noises = [sh(O.synth_noise(Bg, 10 + i, DT)) for i in range(3)]
    alphas = [sh(O.synth_alpha(Bg, 10 + i, DT).reshape(Bg)) for i in range(2)]
    gs.train_iteration(sh(images), sh(labels), noises, alphas, critic_iters=2)
    gs.flush()
    torch.save({"D": gs.D.arena.flat, "G": gs.G.arena.flat}, out % rank)
    torch.distributed.destroy_process_group()


def test_two_rank_dp_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    import sgg_amd  # noqa: F401
    from oracle import sgg_oracle as O
    from oracle.kernels_ref import RefKernels
    from sgg_amd.step import GanStep
    out = str(tmp_path / "rank%d.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["D"], r1["D"]) and torch.equal(r0["G"], r1["G"]), "replicas diverged"
    DT = torch.float64
    Bg, S, V = 4, 32, 11
    gp, dp_ = O.init_params("G", V, S, dtype=DT, perturb=0.1), O.init_params("D", V, S, dtype=DT, perturb=0.1)
    dp_["W"] = dp_["W"] * 25.0
    images, labels, _ = O.synth_batch(Bg, S, V, dtype=DT)
    gs = GanStep(RefKernels(), V, S, Bg, g_state=gp, d_state=dp_, dtype=DT)
    gs.critic_step(images, labels, O.synth_noise(Bg, 0, DT), O.synth_alpha(Bg, 0, DT).reshape(Bg))
    gs.generator_step(images, O.synth_noise(Bg, 1, DT))
    gs.train_iteration(images, labels, [O.synth_noise(Bg, 10 + i, DT) for i in range(3)],
                       [O.synth_alpha(Bg, 10 + i, DT).reshape(Bg) for i in range(2)], critic_iters=2)
    for k, ref in (("D", gs.D.arena.flat), ("G", gs.G.arena.flat)):
        err = float((r0[k] - ref).abs().max())
        assert err < 1e-8, "%s weights: data-parallel vs single process differ by %.3e" % (k, err)
