"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone; nothing in the product path may import oracle/).

CPU restatement of the reference's evaluation, train.py:294-335 (`_recall`, `test`), on top of oracle/sgg_oracle.py:

  per test image: TEST_BATCH_MULTIPLIER sess.run calls, each on TEST_BATCH_SIZE copies of the image (train.py:139-150 repeats every
  test image's entries up to TEST_BATCH_SIZE * TEST_BATCH_MULTIPLIER), each giving
      fake_triples = argmax(G(images), -1)                 [TB, 3]        (:270, :312)
      disc_scores  = mean(D(G(images), images), axis=1)    [TB, 1]        (:272, :315)
  accumulated (:317-319); indices = score_accumulator.argsort() (:321); top 50 / 100 (:322-323); recall = |set(fake[top]) & set(real)| / 50
  resp. 100 (:294-295, :325-326).

`literal=True` follows the code to the letter: score_accumulator has shape [N, 1], argsort works along the LAST axis (length 1), so
`indices` is an [N, 1] array of zeros, `fake_accumulator[indices[:50]]` is sample 0 repeated 50 times ([50, 1, 3] -> squeeze -> [50, 3]).
`literal=False` is the ordering the line is written to mean: ascending mean critic score over the N samples.
The noise of each generator run is an explicit input (the reference draws tf.random_normal inside the graph).
Parity unpinned, as the whole oracle: the reference cannot run here and ships no fixtures.
"""
import numpy as np
import torch

from . import sgg_oracle as O


def recall(fake, real, N):
    """train.py:294-295."""
    return float(len(set(map(tuple, fake)).intersection(set(map(tuple, real))))) / N


def evaluate_image(gp, dp, image, real_triples, noises, literal=False):
    """image [S, S, 3]; noises: list of [rows, 512] tensors, one per generator run (rows samples each).
    Returns dict(tokens [N,3] int64, scores [N] float32, r50, r100)."""
    fake_acc = np.empty((0, 3), dtype=np.float64)
    score_acc = np.empty((0, 1), dtype=np.float64)
    margin = float("inf")
    with torch.no_grad():
        for noise in noises:
            rows = noise.shape[0]
            images = image.unsqueeze(0).expand(rows, -1, -1, -1).contiguous()
            fake_inputs = O.generator_forward(gp, images, noise)                     # [rows, 3, V] raw logits
            fake_triples = O.argmax_tokens(fake_inputs).numpy()                      # tf.argmax(fake_inputs, axis=-1)
            margin = min(margin, O.top2_margin(fake_inputs))                         # (how well the arg-max is resolved: for the tests)
            disc_fake = O.discriminator_forward(dp, fake_inputs, images).numpy()     # [rows, 3, 1]
            disc_scores = np.mean(disc_fake, axis=1)                                 # [rows, 1]
            fake_acc = np.append(fake_acc, fake_triples, axis=0)
            score_acc = np.append(score_acc, disc_scores, axis=0)
    real_acc = np.asarray(real_triples, dtype=np.float64).reshape(-1, 3)
    if literal:
        indices = score_acc.argsort()                                                # [N, 1] zeros: the reference's line 321 as written
        sel50 = np.squeeze(fake_acc[indices[:50]], axis=1)
        sel100 = np.squeeze(fake_acc[indices[:100]], axis=1)
    else:
        indices = score_acc.reshape(-1).argsort(kind="stable")
        sel50, sel100 = fake_acc[indices[:50]], fake_acc[indices[:100]]
    return {"tokens": fake_acc.astype(np.int64), "scores": score_acc.reshape(-1).astype(np.float32),
            "r50": recall(sel50, real_acc, 50.0), "r100": recall(sel100, real_acc, 100.0), "top2_logit_margin": margin}
