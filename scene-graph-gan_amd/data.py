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


def decode_rgb(path):
    """tf.image.decode_jpeg(channels=3) (train.py:169): uint8 [H, W, 3]."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB")))


def _parse_task(task):
    """Worker-process entry point: (path, means, stds, side) -> float32 [side, side, 3]."""
    path, means, stds, side = task
    return parse_image(path, means, stds, side)


class ShuffledStream:
    """Example indices in the order of the reference's training / validation datasets (train.py:176-179):

        Dataset.from_tensor_slices(files, labels).repeat().shuffle(buffer_size = 10 * batch_size)

    i.e. the (once-shuffled, train.py:130) file list walked cyclically for ever, passed through tf.data's rolling shuffle buffer:
    the buffer is filled with the first `buffer` elements of the stream; every draw takes a uniformly random slot of the buffer and
    refills that slot with the stream's next element.  An element therefore never appears earlier than `buffer - 1` positions
    before its place in the cyclic walk, and every element of the walk is emitted exactly once.  TF's own random stream is not
    reproducible outside TF: `seed` selects a NumPy one, fixed per (seed, stream), so a run and its resumed continuation see
    the same order (`batch(it, ...)` is a pure function of `it`)."""

    def __init__(self, n, buffer, seed=0):
        assert n > 0 and buffer > 0
        self.n, self.buffer, self.seed = int(n), int(buffer), seed
        self._reset()

    def _reset(self):
        self.rng = np.random.RandomState(self.seed)
        self.buf = [i % self.n for i in range(self.buffer)]
        self.next_in = self.buffer
        self.emitted = 0

    def draw(self):
        j = int(self.rng.randint(len(self.buf)))
        out = self.buf[j]
        self.buf[j] = self.next_in % self.n
        self.next_in += 1
        self.emitted += 1
        return out

    def take(self, start, count):
        """Elements [start, start + count) of the shuffled stream (rewinds and replays when `start` lies behind the cursor)."""
        if start < self.emitted:
            self._reset()
        self._skip(start - self.emitted)
        return [self.draw() for _ in range(count)]

    def _skip(self, count):
        """Advance the stream by `count` draws without materialising them (a resumed run starts at iteration itr: itr * B * world
        draws).  Same random stream as `draw` (RandomState.randint yields the same values one at a time and in blocks: pinned in
        tests/test_data_eval.py); only the buffer state matters: a slot holds the element put there by the LAST draw that hit it."""
        buf = np.asarray(self.buf, dtype=np.int64)
        while count > 0:
            m = min(count, 1 << 20)
            js = self.rng.randint(len(self.buf), size=m)
            last = np.full(len(self.buf), -1, dtype=np.int64)
            last[js] = np.arange(m, dtype=np.int64)              # repeated indices: the last assignment stays
            hit = last >= 0
            buf[hit] = (self.next_in + last[hit]) % self.n
            self.next_in += m
            self.emitted += m
            count -= m
        self.buf = buf.tolist()

    def batch(self, it, batch_size, rank=0, world=1):
        """Global batch `it` is elements [it * B * world, (it + 1) * B * world) of the stream; rank r takes rows [r * B, (r + 1) * B)."""
        return self.take(it * world * batch_size, world * batch_size)[rank * batch_size:(rank + 1) * batch_size]


class PrefetchLoader:
    """Batches of (images [B, side, side, 3] float32, labels [B, 3] int64) on `device`, produced ahead of the consumer.

    On a HIP device (the product path): worker threads only DECODE (libjpeg releases the GIL); the producer thread packs the
    batch's uint8 images back to back into one of `depth` pinned host buffers, copies them to the device on its own HIP stream
    and launches ONE resize + standardise kernel for the whole batch there (sgg_resize_bilinear_tf1: the TF-1.x arithmetic,
    equal to `resize_bilinear_tf1` within a few ulps), so batch k+1 is decoded, copied and resized while batch k trains.
    On the CPU (tests; `processes=True` uses spawned worker processes instead of threads) the workers run `parse_image`.
    `index_fn(it)` gives the example indices of iteration `it` (the trainer's shard of the shuffled file list); iteration
    order is deterministic."""

    def __init__(self, files, labels, batch_size, index_fn, means, stds, device, num_iterations, start=0, workers=16, depth=2,
                 side=IMAGE_SIDE, processes=False):
        self.files, self.labels = files, np.asarray(labels, dtype=np.int64)
        self.B, self.index_fn, self.means, self.stds, self.side = batch_size, index_fn, means, stds, side
        self.device = torch.device(device)
        self.start, self.stop = start, num_iterations
        self.processes = processes and self.device.type != "cuda"      # on a HIP device the workers only decode: threads
        if self.processes:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            self.pool = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))
        else:
            self.pool = ThreadPoolExecutor(max_workers=workers)
        self.depth = depth
        pin = self.device.type == "cuda"
        self.K = None
        if pin:
            from .lib import HipKernels
            self.K = HipKernels(self.device)
            self.packed = [torch.empty(batch_size * 640 * 480 * 3, dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
            self.meta = [torch.empty(batch_size * 2, dtype=torch.int64, pin_memory=True) for _ in range(depth)]   # offsets | (h, w) pairs
            self.d_means = torch.tensor(np.asarray(means, np.float32), device=self.device)
            self.d_stds = torch.tensor(np.asarray(stds, np.float32), device=self.device)
        self.host = [torch.empty((batch_size, side, side, 3), dtype=torch.float32) for _ in range(depth if not pin else 0)]
        self.free = queue.Queue()
        for i in range(depth):
            self.free.put(i)
        self.ready = queue.Queue(maxsize=depth)
        self.copy_stream = torch.cuda.Stream(device=self.device) if pin else None
        self.error = None
        self._stop = threading.Event()
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _get_free(self):
        while not self._stop.is_set():
            try:
                return self.free.get(timeout=0.1)
            except queue.Empty:
                continue
        return None

    def _put_ready(self, item):
        while not self._stop.is_set():
            try:
                self.ready.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def close(self):
        """Stop the producer (training ended early, or an exception in the consumer): set the stop flag, drain the queue the
        producer may be blocked on, join it and release the decode pool.  Idempotent."""
        self._stop.set()
        try:
            while True:
                self.ready.get_nowait()
        except queue.Empty:
            pass
        if self.thread.is_alive() and threading.current_thread() is not self.thread:
            self.thread.join(timeout=30.0)
        try:
            self.pool.shutdown(wait=False, cancel_futures=True)
        except TypeError:                   # (cancel_futures: Python >= 3.9)
            self.pool.shutdown(wait=False)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _fill(self, dst, j, path):
        dst[j] = torch.from_numpy(parse_image(path, self.means, self.stds, self.side))

    def _produce(self):
        try:
            for it in range(self.start, self.stop):
                idx = self.index_fn(it)
                slot = self._get_free()
                if slot is None:
                    return
                if self.K is not None:
                    if not self._put_ready(self._produce_device(slot, idx)):
                        return
                    continue
                buf = self.host[slot]
                if self.processes:
                    means, stds = np.asarray(self.means, np.float32), np.asarray(self.stds, np.float32)
                    tasks = [(self.files[i], means, stds, self.side) for i in idx]
                    for j, arr in enumerate(self.pool.map(_parse_task, tasks, chunksize=max(1, len(tasks) // 64))):
                        buf[j] = torch.from_numpy(arr)
                else:
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
                if not self._put_ready((slot, dev_images, dev_labels, done)):
                    return
        except BaseException as e:        # surfaced to the consumer: a failed decode must not look like end-of-data
            self.error = e
        finally:
            self._put_ready(None)         # (gives up when close() was called: nobody is waiting then)

    def _produce_device(self, slot, idx):
        """Decode on worker threads, pack, one host-to-device copy, one resize + standardise launch (copy stream)."""
        rgbs = list(self.pool.map(lambda i: decode_rgb(self.files[i]), idx))
        sizes = [a.size for a in rgbs]
        total = int(sum(sizes))
        if self.packed[slot].numel() < total:
            self.packed[slot] = torch.empty(int(total * 1.25), dtype=torch.uint8, pin_memory=True)
        packed, meta = self.packed[slot], self.meta[slot]
        pk = packed.numpy()
        off = 0
        hw = np.empty((len(rgbs), 2), np.int32)
        offs = np.empty(len(rgbs), np.int64)
        for j, a in enumerate(rgbs):
            pk[off:off + a.size] = a.reshape(-1)
            offs[j] = off
            hw[j] = a.shape[:2]
            off += a.size
        meta.numpy()[:len(rgbs)] = offs
        hw_t = torch.from_numpy(hw)
        labels = torch.from_numpy(self.labels[idx])
        with torch.cuda.stream(self.copy_stream):
            d_packed = packed[:total].to(self.device, non_blocking=True)
            d_offs = meta[:len(rgbs)].to(self.device, non_blocking=True)
            d_hw = hw_t.pin_memory().to(self.device, non_blocking=True)
            dev_labels = labels.pin_memory().to(self.device, non_blocking=True)
            dev_images = torch.empty((len(rgbs), self.side, self.side, 3), dtype=torch.float32, device=self.device)
            d_h, d_w = d_hw[:, 0].contiguous(), d_hw[:, 1].contiguous()
            self.K.resize_bilinear_tf1(d_packed, d_offs, d_h, d_w, dev_images, self.d_means, self.d_stds)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        return (slot, dev_images, dev_labels, done)

    def __iter__(self):
        return self

    def __next__(self):
        if self._stop.is_set():
            raise StopIteration
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
