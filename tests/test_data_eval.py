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
from oracle.data_ref import early_stop_ref, resize_loop, shuffled_stream_ref
from sgg_amd import data as D


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
    # consumers on the producer / consumer kernel (pre-split activations), forward-only passes: the prologue variant is ~20 % slower than
    # the DMA-staged kernel, so only the LayerNorms whose apply pass moves enough bytes per consumer FLOP keep the fusion (round 5)
    from sgg_amd.trunk import pc_ln_fusion_pays
    pc = {4: ((64, 112, 112, 64), 64, 128), 5: ((64, 112, 112, 128), 128, 128), 7: ((64, 56, 56, 128), 128, 256), 8: ((64, 56, 56, 256), 256, 256)}
    assert [i for i, a in pc.items() if pc_ln_fusion_pays(*a)] == [4, 5]


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


def test_shuffled_stream_is_tf_data_repeat_shuffle():
    """train.py:176-179: repeat().shuffle(10 * batch) - sgg_amd.data.ShuffledStream against the element-by-element restatement fed
    with the same slot numbers, plus the properties any such buffer has."""
    n, buf, seed = 37, 12, 5
    st = D.ShuffledStream(n, buf, seed)
    got = st.take(0, 400)
    rng = np.random.RandomState(seed)                                               # the stream draws one slot per output
    slots = [int(rng.randint(buf)) for _ in range(400)]
    assert got == shuffled_stream_ref(n, buf, slots)
    # (a) a pure function of the position: replays, resumes and out-of-order requests agree
    assert D.ShuffledStream(n, buf, seed).take(100, 50) == got[100:150]
    assert st.take(30, 10) == got[30:40] and st.take(390, 10) == got[390:400]
    # (b) an element never appears earlier than buffer - 1 positions before its place in the cyclic walk, and nothing is lost or
    # duplicated: after k outputs exactly the first k + buffer inputs minus the buffer's current content have been emitted
    walk_pos = {}
    for pos, e in enumerate(got):
        k = walk_pos.get(e, -1) + 1
        # occurrence number k of element e sits at input position e + k * n
        assert pos >= e + k * n - (buf - 1), (pos, e, k)
        walk_pos[e] = k
    counts = np.bincount(got, minlength=n)
    assert counts.max() - counts.min() <= 1 + buf // n + 1
    # (c) data parallel: the ranks' rows are disjoint slices of one global batch
    a, b = D.ShuffledStream(n, buf, seed), D.ShuffledStream(n, buf, seed)
    g = D.ShuffledStream(n, buf, seed).take(3 * 8, 8)
    assert a.batch(3, 4, 0, 2) == g[:4] and b.batch(3, 4, 1, 2) == g[4:]
    # a buffer of 1 is the plain cyclic walk
    assert D.ShuffledStream(5, 1, 0).take(0, 12) == [i % 5 for i in range(12)]


def test_validation_early_stop_matches_reference_loop():
    from train import ValidationEarlyStop, _str2bool
    cases = [[5, 4, 3, 2], [1, 2, 3, 4], [3, 4, 5, 4, 5, 6, 7], [1, 1, 1, 1, 1], [2, 3, 4, 1, 2, 3, 4, 5], [float("inf"), 1, 2, 3, 4],
             [1, 2, 3, 2.5, 3, 4, 5]]
    for losses in cases:
        es, got = ValidationEarlyStop(3), None
        for i, l in enumerate(losses):
            if es.update(l):
                got = i
                break
        assert got == early_stop_ref(losses, 3), losses
    assert early_stop_ref([1, 2, 3, 4]) == 3 and early_stop_ref([3, 4, 5, 4, 5, 6, 7]) == 6 and early_stop_ref([1, 1, 1, 1]) is None
    # --resume False must be false (the reference's type=bool makes it true, train.py:410)
    assert _str2bool("False") is False and _str2bool("0") is False and _str2bool("true") is True and _str2bool(True) is True
    with pytest.raises(Exception):
        _str2bool("maybe")


def test_recall_oracle_agrees_with_product_ranking():
    """oracle/eval_ref.py restates train.py:317-326 on NumPy arrays exactly as written; train.SceneGraphGAN.recalls must give the same
    numbers in both ordering modes on the same (tokens, scores, real) - here without any network: the accumulators are inputs."""
    from oracle import eval_ref as E
    ev = _Eval()
    rng = np.random.RandomState(3)
    fake = rng.randint(0, 6, size=(128, 3))                     # few distinct triples: the set semantics matter
    scores = rng.randn(128).astype(np.float32)
    scores[10] = scores[20]                                     # a tie
    real = [fake[5].tolist(), fake[77].tolist(), [9, 9, 9]]
    for literal in (False, True):
        score_acc = scores.astype(np.float64).reshape(-1, 1)
        fake_acc = fake.astype(np.float64)
        if literal:
            idx = score_acc.argsort()
            s50, s100 = np.squeeze(fake_acc[idx[:50]], axis=1), np.squeeze(fake_acc[idx[:100]], axis=1)
        else:
            idx = score_acc.reshape(-1).argsort(kind="stable")
            s50, s100 = fake_acc[idx[:50]], fake_acc[idx[:100]]
        exp = (E.recall(s50, np.asarray(real, np.float64), 50.0), E.recall(s100, np.asarray(real, np.float64), 100.0))
        assert ev.recalls(fake, scores, real, reference_literal=literal) == exp


def test_prefetch_loader_close_releases_a_blocked_producer(tmp_path):
    """Training that stops early (exception, early stop, max_iterations below the loader's stop) must not leave the producer thread
    blocked on its bounded queue holding buffers: close() stops it; iterating afterwards ends."""
    files = _write_jpegs(tmp_path, 6)
    loader = D.PrefetchLoader(files, np.zeros((6, 3), np.int64), 2, lambda it: [(2 * it + j) % 6 for j in range(2)], [0, 0, 0], [1, 1, 1], "cpu",
                              1000, workers=2, side=9)
    first = next(loader)
    assert first[0].shape == (2, 9, 9, 3)
    import time
    time.sleep(0.3)                         # the producer now sits in a full `ready` queue / an empty `free` queue
    assert loader.thread.is_alive()
    loader.close()
    assert not loader.thread.is_alive()
    with pytest.raises(StopIteration):
        next(loader)
    loader.close()                          # idempotent
    with D.PrefetchLoader(files, np.zeros((6, 3), np.int64), 2, lambda it: [0, 1], [0, 0, 0], [1, 1, 1], "cpu", 50, workers=2, side=9) as l2:
        next(l2)
    assert not l2.thread.is_alive()
