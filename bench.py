"""bench.py - triples/sec of the WGAN-GP "G+D step" (one critic update + one generator update on one minibatch,
train.py:362-368 with CRITIC_ITERS = 1) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[1]: batch 64 per GPU, 224x224 synthetic images, vocab 1000, length-3 triples
(weak scaling: the global batch is 64*N, sharded by rows of one seeded global draw; the only exchange is the
gradient all-reduce over RCCL).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip table: Peak FP32 (matrix)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # same table: Peak BF16 MFMA, dense
MFMA_F16_PEAK_TFLOPS = 2500.0     # f16 MFMA runs at the bf16 rate (MI355X_MICROARCH.md, Matrix cores table)
SUSTAINED_F16_MFMA_TFLOPS = 1635.0   # measured, random operands (see the roofline note below)
DTYPE_NAME = {0: "f32 (native f32 MFMA)",
              2: "f32 (conv operands scaled per tensor and split into 2 fp16 pieces = 22 significant bits, 3 fp16 MFMAs per "
                 "product, f32 accumulate; error vs fp64 = native f32)",
              6: "f32 (conv operands split into 3 bf16 pieces, 6 bf16 MFMAs per product, f32 accumulate; error vs fp64 = native f32)",
              3: "f32 storage, conv operands split into 2 bf16 pieces (3 bf16 MFMAs per product, f32 accumulate)"}


def conv_flops_per_step(B, S):
    """8*F_fwd - 2*f1: G-conv fwd x2, D-conv fwd x2, one backward (dgrad+wgrad, no image dgrad) each."""
    from sgg_amd.params import CONV_SPECS, same_pads
    h, f_fwd, f1 = S, 0.0, 0.0
    for (i, cin, cout, k, s, _, live) in CONV_SPECS:
        if not live:
            continue
        ho = same_pads(h, k, s)[0]
        f = 2.0 * B * ho * ho * cout * k * k * cin
        f_fwd += f
        if i == 0:
            f1 = f
        h = ho
    return 8 * f_fwd - 2 * f1


def synth_inputs(B_global, S, V, n_draws, rank, world, device):
    """images N(0,1) seed 0, labels randint seed 1, noise N(0,1) seed 2+k, alpha U[0,1) seed 1000+k,
    drawn for the GLOBAL batch on a CPU generator and sliced by rank."""
    B = B_global // world
    sl = slice(rank * B, (rank + 1) * B)
    g = torch.Generator().manual_seed(0)
    images = torch.randn((B_global, S, S, 3), generator=g)[sl].contiguous().to(device)
    g = torch.Generator().manual_seed(1)
    labels = torch.randint(0, V, (B_global, 3), generator=g, dtype=torch.int64)[sl].contiguous().to(device)
    noises, alphas = [], []
    for k in range(n_draws):
        g = torch.Generator().manual_seed(2 + k)
        noises.append(torch.randn((B_global, 512), generator=g)[sl].contiguous().to(device))
        g = torch.Generator().manual_seed(1000 + k)
        alphas.append(torch.rand((B_global,), generator=g)[sl].contiguous().to(device))
    return images, labels, noises, alphas


def cpu_baseline(S, V, rows, threads, steps=2):
    """The CPU oracle (restatement of the reference, oracle/sgg_oracle.py) timed on this box's host cores on a
    bounded sample of the same workload: `rows` rows of the 64-row batch, `steps` full G+D steps (about 10-30 s)."""
    from oracle import sgg_oracle as O
    torch.set_num_threads(threads)
    gp, dp = O.init_params("G", V, S), O.init_params("D", V, S)
    images, labels, onehot = O.synth_batch(rows, S, V)
    d_adam, g_adam = O.new_adam_state(dp), O.new_adam_state(gp)
    t0 = time.time()
    for k in range(steps):
        O.d_step(gp, dp, d_adam, k + 1, images, onehot, O.synth_noise(rows, 2 * k), O.synth_alpha(rows, k))
        O.g_step(gp, dp, g_adam, k + 1, images, O.synth_noise(rows, 2 * k + 1))
    dt = time.time() - t0
    return {"value": rows * steps / dt, "unit": "triples/sec", "cores": threads, "kind": "port",
            "sample": "%d G+D steps on %d of the 64 rows of configs[1] (%dx%d, vocab %d), oracle/sgg_oracle.py fp32 "
                      "(PyTorch CPU restatement of the reference; the reference itself cannot run here), "
                      "%d torch threads, %.1f s" % (steps, rows, S, S, V, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="rows per GPU (configs[1]: 64)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--vocab", type=int, default=1000)
    ap.add_argument("--critic-iters", type=int, default=1)
    ap.add_argument("--cpu-rows", type=int, default=16, help="rows of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--per-shape", action="store_true", help="add per-layer conv timings to the JSON line")
    ap.add_argument("--overlap-streams", action="store_true", help="run the two encoders' forwards on two HIP streams (+3 %%)")
    ap.add_argument("--conv-precision", type=int, default=None, choices=[0, 2, 3, 6],
                    help="conv contraction: 2 = scaled fp16 pieces, 3 products (default, f32-equivalent error), 6 = bf16 pieces, "
                         "6 products (f32-equivalent), 0 = native f32 MFMA, 3 = bf16 pieces, 3 products (within the 1e-4 tolerance)")
    args = ap.parse_args()

    import sgg_amd  # noqa: F401
    from sgg_amd import dp as dpmod
    from sgg_amd.lib import HipKernels
    from sgg_amd.params import init_state_dict
    from sgg_amd.step import GanStep

    rank, world, local = dpmod.init_from_env()
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d): launch N>1 through torch.distributed.run" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda:%d" % local)
    K = HipKernels(dev)
    if args.conv_precision is not None:
        K.conv_precision = args.conv_precision
    B, S, V, CI = args.batch, args.size, args.vocab, args.critic_iters
    reducer = dpmod.GradReducer() if world > 1 else None
    gs = GanStep(K, V, S, B, lam=10.0, g_state=init_state_dict("G", V, S), d_state=init_state_dict("D", V, S), reducer=reducer,
                 overlap_streams=args.overlap_streams)
    total_steps = args.warmup + args.steps
    images, labels, noises, alphas = synth_inputs(B * world, S, V, total_steps * (CI + 1), rank, world, dev)

    def one_step(k):
        base = k * (CI + 1)
        for i in range(CI):
            gs.critic_step(images, labels, noises[base + i], alphas[base + i])
        gs.generator_step(images, noises[base + CI])

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        one_step(k)
    gs.flush()
    barrier()
    if not args.no_kernel_timing:
        K.timing = []
    t0 = time.perf_counter()
    for k in range(args.warmup, total_steps):
        one_step(k)
    gs.flush()
    barrier()
    dt = time.perf_counter() - t0
    timing, K.timing = K.timing, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    d_losses, g_losses = gs.d_losses.cpu().tolist(), gs.g_losses.cpu().tolist()

    if rank == 0:
        value = B * world * args.steps / dt
        out = {
            "metric": "triples/sec (G+D step)", "value": value, "unit": "triples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NAME[K.conv_precision], "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: batch %d per GPU, %dx%d synthetic images, vocab %d, 3-token "
                                   "triples, 1 critic update + 1 generator update per step (WGAN-GP lambda=10, TF-Adam)"
                                   % (B, S, S, V), "global_batch": B * world, "critic_iters": CI,
                       "parallelism": "dp%d" % world},
            "losses": {"disc_cost": d_losses[0], "gp": d_losses[2], "gen_cost": -g_losses[3]},
        }
        flops_step = conv_flops_per_step(B, S) * (CI + 1) / 2.0 if CI == 1 else None
        if flops_step:
            out["conv_tflops_whole_step_per_gpu"] = flops_step * args.steps / dt / 1e12
        if timing:
            per = {}
            for sym, fl, e0, e1 in timing:
                a = per.setdefault(sym, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += fl
                a[2] += e0.elapsed_time(e1) * 1e-3
            dom = max((s for s in per if s.startswith(("conv_gather", "conv_halo"))), key=lambda s: per[s][2])
            n, fl, sec = per[dom]
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(dom)
            nprod = {0: 1, 2: 3, 3: 3, 6: 6}[K.conv_precision]
            peak = MFMA_BF16_PEAK_TFLOPS / nprod if K.conv_precision else MFMA_F32_PEAK_TFLOPS
            note = ("dense %s MFMA peak %.0f TFLOP/s / %d MFMA products per algorithmic f32 product"
                    % ("f16" if K.conv_precision == 2 else "bf16", MFMA_BF16_PEAK_TFLOPS, nprod)
                    if K.conv_precision else "f32 matrix peak")
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": fl / sec / 1e12, "peak": peak, "peak_note": note,
                               "unit": "TFLOP/s", "frac": fl / sec / 1e12 / peak, "traffic": traffic,
                               "mfma_tflops_issued": nprod * fl / sec / 1e12, "vs_native_f32_mfma_peak": fl / sec / 1e12 / MFMA_F32_PEAK_TFLOPS,
                               "launches": n, "avg_launch_ms": 1e3 * sec / n, "flop_per_launch": fl / n,
                               "share_of_step_time": sec / dt}
            if K.conv_precision in (2, 3):
                # scripts/ubench/mfma16_peak.hip (profiles/r01_ubench_mfma16_peak.log): a register-only loop of
                # v_mfma_f32_32x32x16_f16 on random operands sustains 1.56-1.71 PFLOP/s (the chip clocks down to 1.5-1.7 GHz
                # under matrix load; 2.46 PFLOP/s only with all-zero operands) -> / 3 products
                out["roofline"]["sustained_mfma_peak_measured"] = SUSTAINED_F16_MFMA_TFLOPS / nprod
                out["roofline"]["frac_of_sustained"] = fl / sec / 1e12 / (SUSTAINED_F16_MFMA_TFLOPS / nprod)
            out["kernel_time_s"] = {s: round(v[2], 4) for s, v in sorted(per.items(), key=lambda kv: -kv[1][2])}
            out["kernel_tflops"] = {s: round(v[1] / v[2] / 1e12, 2) for s, v in per.items() if v[2] > 0}
            if args.per_shape:     # per (kernel, FLOPs per launch) = per layer and direction
                shp = {}
                for sym, fl, e0, e1 in timing:
                    a = shp.setdefault("%s @ %.1f GF" % (sym, fl / 1e9), [0, 0.0])
                    a[0] += 1
                    a[1] += e0.elapsed_time(e1) * 1e-3
                out["per_shape"] = {k: {"launches": v[0], "ms_per_step": round(1e3 * v[1] / args.steps, 3),
                                        "tflops": round(float(k.split("@")[1].split()[0]) * 1e9 * v[0] / v[1] / 1e12, 1)}
                                    for k, v in sorted(shp.items(), key=lambda kv: -kv[1][1])}
        if world == 1 and args.cpu_rows > 0:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(S, V, args.cpu_rows, min(ncpu, 16))   # a 1-GPU box grants 16 host cores
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
