"""Throughput of the real-data input pipeline (sgg_amd/data.py: JPEG decode -> TF-1.x resize 221x221 -> standardise -> pinned
double buffer -> device) on generated Visual-Genome-sized JPEGs:  python scripts/loader_rate.py [n_images] [workers]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
import sgg_amd  # noqa: F401
from sgg_amd.data import PrefetchLoader

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    procs = (sys.argv[3] == "proc") if len(sys.argv) > 3 else True
    B, iters = 64, 26
    d = tempfile.mkdtemp()
    rng = np.random.RandomState(0)
    files = []
    for i in range(n):
        arr = (rng.rand(375, 500, 3) * 255).astype(np.uint8)          # a typical Visual Genome frame
        arr = (0.5 * arr + 0.5 * np.roll(arr, 1, axis=0)).astype(np.uint8)
        p = os.path.join(d, "im%04d.jpg" % i)
        Image.fromarray(arr).save(p, quality=90)
        files.append(p)
    labels = rng.randint(0, 1000, (n, 3))
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    loader = PrefetchLoader(files, labels, B, lambda it: [(it * B + j) % n for j in range(B)], [120.0, 115.0, 100.0], [60.0, 58.0, 61.0],
                            dev, iters, workers=workers, processes=procs)
    next(loader); next(loader)                                         # first batches: worker start-up
    t0 = time.time()
    k = 0
    for images, labs in loader:
        k += 1
    if dev != "cpu":
        torch.cuda.synchronize()
    dt = time.time() - t0
    print("prefetching loader: %d batches of %d in %.2f s = %.0f images/s (%d decode %s, %d host cores, device %s)"
          % (k, B, dt, k * B / dt, workers, "threads + device-side resize" if dev != "cpu" else ("processes" if procs else "threads"), len(os.sched_getaffinity(0)), dev))


if __name__ == "__main__":      # (worker processes are spawned: they re-import this module)
    main()
