"""CPU: rows f1 / f3 of SURVEY.md 8 with real oracles.

f1: sgg_amd.data.resize_bilinear_tf1 against a plain-loop restatement of TF-1.x tf.image.resize_images (bilinear,
    align_corners=False, legacy grid src = dst * in/out, no antialiasing; train.py:171) on odd sizes, up- and down-scaling;
    the prefetching loader delivers exactly the synchronous pipeline's batches.
f3: R@50 / R@100 known answers: ordering by mean critic score, the set semantics of train.py:294-295, and the literal
    behaviour of the reference's [N,1] argsort (train.py:315-323).
"""
import os

import numpy as np
import pytest
import torch

import sgg_amd  # noqa: F401
from sgg_amd import data as D


def resize_loop(img, oh, ow):
    """tf.image.resize_images(img, [oh, ow]) of TF 1.x, one output pixel at a time."""
    H, W, C = img.shape
    out = np.zeros((oh, ow, C), dtype=np.float32)
    sy, sx = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)
    for y in range(oh):
        fy = np.float32(y) * sy
        y0 = int(np.floor(fy)); y1 = min(y0 + 1, H - 1); wy = np.float32(fy - y0)
        for x in range(ow):
            fx = np.float32(x) * sx
            x0 = int(np.floor(fx)); x1 = min(x0 + 1, W - 1); wx = np.float32(fx - x0)
            top = img[y0, x0] + (img[y0, x1] - img[y0, x0]) * wx
            bot = img[y1, x0] + (img[y1, x1] - img[y1, x0]) * wx
            out[y, x] = top + (bot - top) * wy
    return out


@pytest.mark.parametrize("shape,out", [((37, 53), (21, 21)), ((13, 9), (21, 17)), ((50, 75), (22, 22)), ((5, 5), (5, 5)), ((1, 7), (3, 3))])
def test_resize_matches_tf1_loop(shape, out):
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=shape + (3,)).astype(np.uint8)
    got = D.resize_bilinear_tf1(img, *out)
    exp = resize_loop(img.astype(np.float32), *out)
    assert got.dtype == np.float32 and got.shape == out + (3,)
    assert np.abs(got - exp).max() <= 1e-4


def test_resize_known_answers():
    # 2x1 -> 4x1: legacy grid src = dst * 0.5 -> rows 0, 0.5, 1, 1.5(clamped: lower 1, upper 1) -> 10, 15, 20, 20
    img = np.array([[[10.0]], [[20.0]]], dtype=np.float32)
    assert np.allclose(D.resize_bilinear_tf1(img, 4, 1)[:, 0, 0], [10, 15, 20, 20])
    # 4 -> 2 keeps samples 0 and 2 exactly (no antialiasing; PIL's BILINEAR would average neighbours)
    img = np.array([[[1.0], [2.0], [4.0], [8.0]]], dtype=np.float32)
    assert np.allclose(D.resize_bilinear_tf1(img, 1, 2)[0, :, 0], [1, 4])
    # half-pixel-centre sampling (TF2 / PIL) would give 1.5 and 6.0 here
    assert not np.allclose(D.resize_bilinear_tf1(img, 1, 2)[0, :, 0], [1.5, 6.0])


def _write_jpegs(tmp_path, n):
    from PIL import Image
    rng = np.random.RandomState(1)
    files = []
    for i in range(n):
        h, w = 40 + 3 * (i % 7), 60 + 5 * (i % 5)
        arr = (rng.rand(h, w, 3) * 255).astype(np.uint8)
        p = os.path.join(str(tmp_path), "im%03d.jpg" % i)
        Image.fromarray(arr).save(p, quality=92)
        files.append(p)
    return files


def test_prefetch_loader_equals_synchronous_pipeline(tmp_path):
    files = _write_jpegs(tmp_path, 24)
    labels = np.arange(24 * 3).reshape(24, 3)
    means, stds = np.array([120.0, 115.0, 100.0], np.float32), np.array([60.0, 58.0, 61.0], np.float32)
    B = 8
    index_fn = lambda it: [(it * B + j * 5) % 24 for j in range(B)]
    loader = D.PrefetchLoader(files, labels, B, index_fn, means, stds, "cpu", 5, start=1, workers=4, side=33)
    got = list(loader)
    assert len(got) == 4
    for k, (images, labs) in enumerate(got):
        idx = index_fn(k + 1)
        exp = np.stack([D.parse_image(files[i], means, stds, 33) for i in idx])
        assert images.shape == (B, 33, 33, 3) and images.dtype == torch.float32
        assert np.array_equal(images.numpy(), exp)
        assert np.array_equal(labs.numpy(), labels[idx])
    x = D.parse_image(files[0], means, stds)
    assert x.shape == (221, 221, 3) and np.isfinite(x).all()


def test_prefetch_loader_with_worker_processes(tmp_path):
    files = _write_jpegs(tmp_path, 6)
    labels = np.arange(18).reshape(6, 3)
    means, stds = np.array([120.0, 115.0, 100.0], np.float32), np.array([60.0, 58.0, 61.0], np.float32)
    loader = D.PrefetchLoader(files, labels, 3, lambda it: [(2 * it + j) % 6 for j in range(3)], means, stds, "cpu", 2, workers=2, side=17,
                              processes=True)
    got = list(loader)
    assert len(got) == 2
    for k, (images, labs) in enumerate(got):
        idx = [(2 * k + j) % 6 for j in range(3)]
        assert np.array_equal(images.numpy(), np.stack([D.parse_image(files[i], means, stds, 17) for i in idx]))
        assert np.array_equal(labs.numpy(), labels[idx])


def test_prefetch_loader_surfaces_decode_errors(tmp_path):
    files = _write_jpegs(tmp_path, 4) + [os.path.join(str(tmp_path), "missing.jpg")]
    loader = D.PrefetchLoader(files, np.zeros((5, 3), np.int64), 5, lambda it: list(range(5)), [0, 0, 0], [1, 1, 1], "cpu", 1, workers=2, side=8)
    with pytest.raises(Exception):
        list(loader)


class _Eval:
    """The evaluation helpers of train.SceneGraphGAN without constructing the networks."""
    from train import SceneGraphGAN as _S
    _recall, _rank, recalls = _S._recall, staticmethod(_S._rank), _S.recalls


def test_recall_known_answers():
    ev = _Eval()
    N = 128
    fake = np.stack([np.arange(N), np.arange(N) + 1000, np.arange(N) + 2000], axis=1)        # all distinct
    scores = np.arange(N)[::-1].astype(np.float64)          # sample N-1 has the LOWEST score -> ranked first
    # true triples: two of them among the 50 lowest-scored samples, one more among the next 50, one never generated
    real = [fake[N - 1].tolist(), fake[N - 50].tolist(), fake[N - 51].tolist(), [7, 7, 7]]
    r50, r100 = ev.recalls(fake, scores, real)
    assert r50 == 2 / 50.0 and r100 == 3 / 100.0
    # ordering matters: with the scores negated the order is 0..N-1: none of them is in the first 50, samples 77 and 78 in the first 100
    r50b, r100b = ev.recalls(fake, -scores, real)
    assert r50b == 0.0 and r100b == 2 / 100.0
    # set semantics: duplicate generated triples collapse, duplicated true triples too; denominators stay 50 / 100
    fake_dup = np.tile(fake[:1], (N, 1))
    r50c, r100c = ev.recalls(fake_dup, scores, [fake[0].tolist(), fake[0].tolist()])
    assert r50c == 1 / 50.0 and r100c == 1 / 100.0
    # ties keep sample order (stable sort)
    assert list(ev._rank([1.0, 0.0, 1.0, 0.0])) == [1, 3, 0, 2]


def test_recall_reference_literal_quirk():
    """train.py:315-323 argsorts an [N,1] array: every selected index is 0."""
    ev = _Eval()
    fake = np.stack([np.arange(120), np.arange(120), np.arange(120)], axis=1)
    scores = np.random.RandomState(0).randn(120)
    s2 = scores.reshape(-1, 1)
    assert np.array_equal(s2.argsort()[:50].reshape(-1), np.zeros(50, dtype=np.int64))       # what NumPy does with [N,1]
    assert np.array_equal(ev._rank(scores, reference_literal=True)[:50], np.zeros(50, dtype=np.int64))
    assert ev.recalls(fake, scores, [fake[0].tolist()], reference_literal=True) == (1 / 50.0, 1 / 100.0)
    assert ev.recalls(fake, scores, [fake[5].tolist()], reference_literal=True) == (0.0, 0.0)


def test_ln_fusion_cost_model_decisions_at_configs1():
    """sgg_amd/trunk.py: ln_fusion_pays - which LayerNorms the default schedule hands to their consumer's patch staging at batch 64 /
    224x224 (measured overheads, DESIGN.md "The LN prologue"): forward-only passes LN0 .. LN8; passes with a
    backward LN0, LN1, LN4, LN5, LN6 (the trunk additionally asks the kernel set whether the consumer has the prologue)."""
    import sgg_amd  # noqa: F401
    from sgg_amd.trunk import ln_fusion_pays
    from sgg_amd.params import CONV_SPECS, same_pads
    live = [c for c in CONV_SPECS if c[6]]
    h, got = 224, {}
    for idx, (i, cin, cout, k, s, has_ln, _) in enumerate(live[:-1]):
        h = same_pads(h, k, s)[0]
        got[i] = ln_fusion_pays((64, h, h, cout), live[idx + 1][2])
    assert [i for i, v in got.items() if v[0]] == [0, 1, 2, 3, 4, 5, 6, 7, 8]
    assert [i for i, v in got.items() if v[1]] == [0, 1, 4, 5, 6]
    assert ln_fusion_pays((8, 64, 64, 32), 32) == (False, False)            # small tensors: the flat overhead never pays


def test_canvas_plan_for_odd_image_sizes():
    """sgg_amd/trunk.py: plan_canvas - the reference's 221 x 221 images (train.py:171; maps 221, 111, 56, 28, 14) sit at offsets
    3 / 1 / 0 of even canvases 224 / 112 / 56 so that the canvas convolutions' own SAME padding coincides with the true one; sizes
    that tile already, or whose canvas would exceed 1.3x the image, keep the plain grid."""
    import sgg_amd  # noqa: F401
    from sgg_amd.trunk import plan_canvas
    plan = plan_canvas(221)
    assert plan is not None and len(plan) == 12
    assert plan[0][:3] == (224, 3, 221) and plan[0][3:] == (224, 3, 221)          # conv1_1: 3x3 stride 1 keeps canvas and offset
    assert plan[2] == (224, 3, 221, 112, 1, 111)                                   # conv1_3: 5x5 stride 2, o_in = 2 * o_out + 1 (odd input)
    assert plan[7] == (112, 1, 111, 56, 0, 56)                                     # conv2_5: odd 111 -> 56 at offset 0
    assert plan[-1][3:] == (14, 0, 14)
    assert plan_canvas(224) is None and plan_canvas(64) is None and plan_canvas(448) is None
    assert plan_canvas(37) is None                                                 # canvas 48: 1.68x the pixels
    p29 = plan_canvas(29)
    assert p29[0][:3] == (32, 3, 29) and p29[2][3:] == (16, 1, 15) and p29[7][3:] == (8, 0, 8)
