"""Host input pipeline of the real-data path: JPEG -> tf.image.resize_images(221x221) -> standardise, and a prefetching
loader that keeps the GPU fed.

Reference: train.py:167-173 (`_parseFunction`: tf.read_file, tf.image.decode_jpeg(channels=3),
tf.image.resize_images(image, [221, 221]), (x - mean) / std, tf.one_hot) and train.py:175-190 (`map_and_batch` with
(CRITIC_ITERS + 1) * 4 parallel batches + prefetch).

`resize_bilinear_tf1` restates TF 1.x `resize_images` with its defaults (method=BILINEAR, align_corners=False), i.e. the
LEGACY sampling grid: src = dst * (in / out) with NO half-pixel offset and NO antialiasing when shrinking -
lower = floor(src), upper = min(lower + 1, in - 1), linear interpolation in float32, rows then columns.  (PIL's BILINEAR
uses pixel centres and widens its filter support when downscaling, so it is a different function.)
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

IMAGE_SIDE = 221                                    # train.py:171


def _axis_weights(in_size, out_size):
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.arange(out_size, dtype=np.float32) * scale
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, in_size - 1)
    return lo, hi, (src - lo.astype(np.float32)).astype(np.float32)


def resize_bilinear_tf1(img, out_h=IMAGE_SIDE, out_w=IMAGE_SIDE):
    """img: [H, W, C] uint8 or float array -> float32 [out_h, out_w, C] (tf.image.resize_images defaults, TF 1.x)."""
    x = np.asarray(img, dtype=np.float32)
    H, W = x.shape[0], x.shape[1]
    y0, y1, fy = _axis_weights(H, out_h)
    x0, x1, fx = _axis_weights(W, out_w)
    top = x[y0]                                     # [out_h, W, C]
    bot = x[y1]
    tl, tr = top[:, x0], top[:, x1]
    bl, br = bot[:, x0], bot[:, x1]
    fxb = fx[None, :, None]
    t = tl + (tr - tl) * fxb
    b = bl + (br - bl) * fxb
    return t + (b - t) * fy[:, None, None]


def parse_image(path, means, stds, side=IMAGE_SIDE):
    """train.py:167-172: decode (3 channels) -> resize_images([221, 221]) -> (x - mean) / std; float32 [side, side, 3]."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    x = resize_bilinear_tf1(rgb, side, side)
    return (x - np.asarray(means, dtype=np.float32)) / np.asarray(stds, dtype=np.float32)


class PrefetchLoader:
    """Batches of (images [B, side, side, 3] float32, labels [B, 3] int64) on `device`, produced ahead of the consumer.

    Worker threads decode / resize / standardise single images (PIL and NumPy release the GIL in their inner loops); a
    producer thread assembles each batch into one of `depth` pinned host buffers and issues the host-to-device copy on its
    own HIP stream, so batch k+1 is decoded and copied while batch k trains.  `index_fn(it)` gives the example indices of
    iteration `it` (the trainer's shard of the shuffled file list); iteration order is deterministic."""

    def __init__(self, files, labels, batch_size, index_fn, means, stds, device, num_iterations, start=0, workers=16, depth=2,
                 side=IMAGE_SIDE):
        self.files, self.labels = files, np.asarray(labels, dtype=np.int64)
        self.B, self.index_fn, self.means, self.stds, self.side = batch_size, index_fn, means, stds, side
        self.device = torch.device(device)
        self.start, self.stop = start, num_iterations
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.depth = depth
        pin = self.device.type == "cuda"
        self.host = [torch.empty((batch_size, side, side, 3), dtype=torch.float32, pin_memory=pin) for _ in range(depth)]
        self.free = queue.Queue()
        for i in range(depth):
            self.free.put(i)
        self.ready = queue.Queue(maxsize=depth)
        self.copy_stream = torch.cuda.Stream(device=self.device) if pin else None
        self.error = None
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _fill(self, dst, j, path):
        dst[j] = torch.from_numpy(parse_image(path, self.means, self.stds, self.side))

    def _produce(self):
        try:
            for it in range(self.start, self.stop):
                idx = self.index_fn(it)
                slot = self.free.get()
                buf = self.host[slot]
                list(self.pool.map(lambda jp: self._fill(buf, jp[0], self.files[jp[1]]), enumerate(idx)))
                labels = torch.from_numpy(self.labels[idx])
                if self.copy_stream is not None:
                    with torch.cuda.stream(self.copy_stream):
                        dev_images = buf.to(self.device, non_blocking=True)
                        dev_labels = labels.pin_memory().to(self.device, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record(self.copy_stream)
                else:
                    dev_images, dev_labels, done = buf.clone(), labels, None
                self.ready.put((slot, dev_images, dev_labels, done))
        except BaseException as e:        # surfaced to the consumer: a failed decode must not look like end-of-data
            self.error = e
        finally:
            self.ready.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.ready.get()
        if item is None:
            self.pool.shutdown(wait=False)
            if self.error is not None:
                raise self.error
            raise StopIteration
        slot, images, labels, done = item
        if done is not None:
            torch.cuda.current_stream(self.device).wait_event(done)     # the compute stream waits for the copy, the host does not
            done.synchronize()                                          # the pinned buffer may be refilled after the copy
            images.record_stream(torch.cuda.current_stream(self.device))
            labels.record_stream(torch.cuda.current_stream(self.device))
        self.free.put(slot)
        return images, labels
