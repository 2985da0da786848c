"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone; nothing in the product path may import oracle/).

CPU restatement of the reference's INPUT semantics, one element at a time:

  resize_loop          tf.image.resize_images(image, [oh, ow]) of TF 1.x with its defaults (train.py:171): bilinear,
                       align_corners=False, the LEGACY sampling grid src = dst * (in / out) (no half-pixel offset, no antialiasing),
                       lower = floor(src), upper = min(lower + 1, in - 1), float32 arithmetic.  TF itself is an un-vendored,
                       un-pinned dependency of the reference; this restates its published ResizeBilinear kernel
                       (compute_interpolation_weights + the lerp order top/bottom then vertical).
  parse_ref            train.py:167-172 `_parseFunction`: decode -> resize -> (x - mean) / std.
  shuffled_stream_ref  train.py:176-179: Dataset.repeat().shuffle(buffer_size): tf.data's ShuffleDataset fills a buffer with the
                       first `buffer` elements, then per output picks a uniformly random slot and refills it with the next input.
                       (Parity unpinned: TF's random stream is not reproducible here; the restatement takes its random slot
                       numbers as an explicit input, as the rest of the oracle does with every source of randomness.)
  early_stop_ref       train.py:358-384: the validation-loss convergence test, as a function of the loss sequence.
"""
import numpy as np


def resize_loop(img, oh, ow):
    H, W, C = img.shape
    img = np.asarray(img, dtype=np.float32)
    out = np.zeros((oh, ow, C), dtype=np.float32)
    sy, sx = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)
    for y in range(oh):
        fy = np.float32(y) * sy
        y0 = int(np.floor(fy)); y1 = min(y0 + 1, H - 1); wy = np.float32(fy - np.float32(y0))
        for x in range(ow):
            fx = np.float32(x) * sx
            x0 = int(np.floor(fx)); x1 = min(x0 + 1, W - 1); wx = np.float32(fx - np.float32(x0))
            top = img[y0, x0] + (img[y0, x1] - img[y0, x0]) * wx
            bot = img[y1, x0] + (img[y1, x1] - img[y1, x0]) * wx
            out[y, x] = top + (bot - top) * wy
    return out


def parse_ref(rgb_u8, means, stds, side=221):
    """rgb_u8: decoded image [H, W, 3] uint8 (tf.image.decode_jpeg(channels=3), train.py:169)."""
    x = resize_loop(rgb_u8.astype(np.float32), side, side)
    return (x - np.asarray(means, np.float32)) / np.asarray(stds, np.float32)


def shuffled_stream_ref(n, buffer, slots):
    """Outputs of range(n) repeated for ever through a shuffle buffer of `buffer` elements; slots[k] = the random slot index
    (0 <= slots[k] < buffer) the k-th output is taken from."""
    buf = [i % n for i in range(buffer)]
    nxt, out = buffer, []
    for j in slots:
        out.append(buf[j])
        buf[j] = nxt % n
        nxt += 1
    return out


def early_stop_ref(losses, patience=3):
    """Index of the validation at which the loop of train.py:358-384 breaks (None: it never does)."""
    convergence_count, last_loss = 0, float("inf")
    for i, loss in enumerate(losses):
        if last_loss < loss:
            convergence_count += 1
        else:
            convergence_count = 0
        if convergence_count == patience:
            return i
        last_loss = loss
    return None
