"""One rank of tests/test_dp_rccl_gpu.py (started by `python -m torch.distributed.run`, one process per GPU, before anything in this
process has touched a GPU): a data-parallel critic update + generator update with HipKernels on cuda:LOCAL_RANK, gradients
all-reduced by RCCL (torch.distributed backend "nccl") through sgg_amd.dp.GradReducer on the product's multi-stream schedule.
Writes the all-reduced gradient SUMS and the weights after Adam to <prefix><rank>.pt.  Not a test module (no test_ prefix)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    out = sys.argv[1]
    import sgg_amd  # noqa: F401
    from oracle import sgg_oracle as O          # (seeded inputs and initial weights only: the checker's helpers, as in every -m gpu test)
    from sgg_amd import dp
    from sgg_amd.lib import HipKernels
    from sgg_amd.step import GanStep
    rank, world, local = dp.init_from_env(backend=os.environ.get("SGG_DP_BACKEND", "nccl"))     # (gloo: the 1-GPU rehearsal of the test)
    dev = torch.device("cuda:%d" % local)
    K = HipKernels(dev)
    Bg, S, V = 4 * world, 64, 50
    gp, dp_ = O.init_params("G", V, S, perturb=0.05), O.init_params("D", V, S, perturb=0.05)
    dp_["W"] = dp_["W"] * 25.0
    images, labels, _ = O.synth_batch(Bg, S, V)
    noise0, noise1, alpha = O.synth_noise(Bg, 0), O.synth_noise(Bg, 1), O.synth_alpha(Bg, 0).reshape(Bg)
    sh = lambda t: dp.shard_rows(t, rank, world).contiguous().to(dev)
    reducer = dp.GradReducer()
    gs = GanStep(K, V, S, Bg // world, lam=10.0, g_state=gp, d_state=dp_, reducer=reducer, overlap_streams=True)
    img = sh(images)
    gs.critic_step(img, sh(labels), sh(noise0), sh(alpha))
    gs.generator_step(img, sh(noise1))
    gs.flush()
    torch.cuda.synchronize(dev)
    torch.save({"world": world, "backend": torch.distributed.get_backend(),
                "D.gradsum": {k: v.cpu() for k, v in gs.D.grads.items()}, "G.gradsum": {k: v.cpu() for k, v in gs.G.grads.items()},
                "D.weights": gs.D.arena.flat.cpu(), "G.weights": gs.G.arena.flat.cpu(),
                "losses": torch.cat([gs.d_losses, gs.g_losses]).cpu()}, "%s%d.pt" % (out, rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
