"""-m gpu: rows f1 (input pipeline) and f3 (evaluation) of SURVEY.md 8 against their oracles, through the product path
(HipKernels -> libsgg_hip.so; train.SceneGraphGAN).

f1  sgg_resize_bilinear_tf1 (device) vs oracle/data_ref.resize_loop (train.py:171-172): odd sizes, up- and down-scaling, a 1-pixel-
    wide image, a batch of different sizes packed back to back; PrefetchLoader(device="cuda") - packed offsets, int32 sizes, pinned
    slot reallocation above 640x480, grayscale JPEGs, slot reuse - vs the oracle's parse_ref on the decoded pixels.
f3  SceneGraphGAN.test() vs oracle/eval_ref.evaluate_image (train.py:297-335) on the same weights, images and noise: identical
    tokens, scores within 1e-4, EQUAL R@50 / R@100, in both ordering modes.
    train()'s validation-loss early stop (train.py:375-384) and test-at-end (:387-388).
"""
import json
import os

import numpy as np
import pytest
import torch

import sgg_amd  # noqa: F401
from oracle import data_ref as DR
from oracle import eval_ref as ER
from oracle import sgg_oracle as O
from sgg_amd import data as D

pytestmark = pytest.mark.gpu


def _device_resize(hip, imgs, oh, ow, means, stds):
    sizes = [a.size for a in imgs]
    packed = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs])).cuda()
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]), dtype=torch.int64).cuda()
    h = torch.tensor([a.shape[0] for a in imgs], dtype=torch.int32).cuda()
    w = torch.tensor([a.shape[1] for a in imgs], dtype=torch.int32).cuda()
    out = torch.full((len(imgs), oh, ow, 3), float("nan"), device="cuda")
    hip.resize_bilinear_tf1(packed, offs, h, w, out, torch.tensor(means, dtype=torch.float32).cuda(), torch.tensor(stds, dtype=torch.float32).cuda())
    return out.cpu().numpy()


@pytest.mark.parametrize("out", [(21, 21), (17, 33), (221, 221)], ids=["21x21", "17x33", "221x221"])
def test_device_resize_matches_tf1_loop_oracle(hip, out):
    rng = np.random.RandomState(0)
    shapes = [(37, 53), (13, 9), (50, 75), (5, 5), (1, 7), (40, 1), (300, 260), (221, 221)]      # odd, up, down, 1 pixel wide / high, identity
    if out == (221, 221):
        shapes = [(37, 53), (1, 7), (40, 1), (250, 300)]          # (the plain loop oracle at 221x221 takes ~1 s per image)
    imgs = [rng.randint(0, 256, size=s + (3,)).astype(np.uint8) for s in shapes]
    means, stds = [119.6, 115.1, 106.1], [30.4, 30.5, 36.7]
    got = _device_resize(hip, imgs, out[0], out[1], means, stds)
    assert np.isfinite(got).all()
    worst = 0.0
    for b, im in enumerate(imgs):
        exp = DR.parse_ref(im, means, stds, side=out[0]) if out[0] == out[1] else \
            (DR.resize_loop(im.astype(np.float32), *out) - np.asarray(means, np.float32)) / np.asarray(stds, np.float32)
        worst = max(worst, float(np.abs(got[b] - exp).max()))
    print("device resize vs loop oracle: max |d| = %.3e" % worst)
    assert worst <= 1e-4
    # raw interpolation (means 0, stds 1): the values themselves, on the uint8 scale
    got = _device_resize(hip, imgs[:2], out[0], out[1], [0, 0, 0], [1, 1, 1])
    for b in range(2):
        assert float(np.abs(got[b] - DR.resize_loop(imgs[b].astype(np.float32), *out)).max()) <= 1e-4


def _write_images(tmp_path, specs):
    from PIL import Image
    rng = np.random.RandomState(7)
    files = []
    for i, (h, w, mode) in enumerate(specs):
        arr = (rng.rand(h, w, 3) * 255).astype(np.uint8) if mode == "RGB" else (rng.rand(h, w) * 255).astype(np.uint8)
        p = os.path.join(str(tmp_path), "im%03d.jpg" % i)
        Image.fromarray(arr, mode=mode).save(p, quality=90)
        files.append(p)
    return files


def test_device_prefetch_loader_matches_oracle_pipeline(tmp_path):
    """The loader path real-data training takes on the GPU: decode threads -> packed uint8 batch in a pinned slot (reallocated when a
    batch exceeds 640x480 pixels per image) -> copy stream -> ONE resize + standardise launch; more iterations than slots."""
    specs = [(48, 64, "RGB"), (33, 21, "RGB"), (60, 45, "L"), (700, 520, "RGB"), (52, 52, "RGB"), (20, 90, "L"), (64, 48, "RGB"), (31, 31, "RGB")]
    files = _write_images(tmp_path, specs)
    labels = np.arange(len(files) * 3).reshape(-1, 3)
    means, stds = np.array([119.6, 115.1, 106.1], np.float32), np.array([30.4, 30.5, 36.7], np.float32)
    B, side, n_it = 3, 37, 7
    index_fn = lambda it: [(3 * it + 2 * j) % len(files) for j in range(B)]
    loader = D.PrefetchLoader(files, labels, B, index_fn, means, stds, "cuda", n_it, workers=3, side=side, depth=2)
    small = loader.packed[0].numel()
    loader.packed = [torch.empty(4096, dtype=torch.uint8, pin_memory=True) for _ in range(2)]      # force the reallocation path on every slot
    got = [(im.clone(), lab.clone()) for im, lab in loader]
    assert len(got) == n_it and small == B * 640 * 480 * 3
    worst = 0.0
    for it, (images, labs) in enumerate(got):
        idx = index_fn(it)
        assert images.is_cuda and tuple(images.shape) == (B, side, side, 3) and labs.dtype == torch.int64
        assert np.array_equal(labs.cpu().numpy(), labels[idx])
        for j, i in enumerate(idx):
            exp = DR.parse_ref(D.decode_rgb(files[i]), means, stds, side=side)
            worst = max(worst, float(np.abs(images[j].cpu().numpy() - exp).max()))
            host = D.parse_image(files[i], means, stds, side)           # the product's synchronous host pipeline: a few ulps
            assert float(np.abs(images[j].cpu().numpy() - host).max()) <= 5e-5
    print("device loader vs oracle pipeline: max |d| = %.3e" % worst)
    assert worst <= 1e-4
    loader.close()


def _gan(tmp_path, B, S, V, **kw):
    import train as T
    return T.SceneGraphGAN(str(tmp_path / "ck"), str(tmp_path / "logs"), None, None, None, None, None, critic_iters=1, batch_size=B,
                           lambda_=10, resume=False, synthetic=(B, S, V), **kw)


def test_evaluation_matches_oracle(tmp_path):
    """SceneGraphGAN.test() (8 x TEST_BATCH_SIZE samples per image, mean critic score, ordering, set recall) against the oracle's
    restatement of train.py:297-335 on the same weights, images and noise."""
    B, S, V = 32, 64, 50                       # TEST_BATCH_SIZE 16 -> 128 samples per image: R@50 and R@100 select different sets
    gan = _gan(tmp_path, B, S, V)
    images0, _ = gan._next_batch(0)
    gan._constructOps(images0)
    gan.train(max_iterations=2, log_every=1000, test_at_end=False)         # trained weights (two Adam steps), then evaluate
    gp, dp = gan.g.state_dict(full_names=False), gan.d.state_dict(full_names=False)
    n_samples = gan.TEST_BATCH_MULTIPLIER * gan.TEST_BATCH_SIZE
    assert n_samples == 128
    g = torch.Generator().manual_seed(4242)
    imgs = [torch.randn((S, S, 3), generator=g) for _ in range(2)]
    # the noise stream test() draws (its generator is seeded from gan.seed): TEST_BATCH_MULTIPLIER runs of TEST_BATCH_SIZE rows per
    # image (train.py:311-318), one [TEST_BATCH_SIZE, 512] draw per run, image after image - at that batch size, on the weights
    # trained at batch B
    gen = torch.Generator().manual_seed(gan.seed + 123)
    TB, passes = gan.TEST_BATCH_SIZE, gan.TEST_BATCH_MULTIPLIER
    assert TB == B // 2 and passes == 8
    noises = [[torch.randn((TB, 512), generator=gen) for _ in range(passes)] for _ in imgs]
    # true triples: taken from what the ORACLE generates (so the recalls are not trivially zero), plus never-generated ones
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    pre = [ER.evaluate_image(gp, dp, im, [[0, 0, 0]], ns) for im, ns in zip(imgs, noises)]
    items = []
    for im, p in zip(imgs, pre):
        order = np.argsort(p["scores"], kind="stable")
        real = [p["tokens"][order[0]].tolist(), p["tokens"][order[60]].tolist(), p["tokens"][order[-1]].tolist(), [V - 1, V - 1, V - 1]]
        items.append((im, real))
    for literal in (False, True):
        (r50, r100), details = gan.test(items=items, out_path=str(tmp_path / "recalls.txt"), reference_literal=literal, return_details=True)
        exp = [ER.evaluate_image(gp, dp, im, real, ns, literal=literal) for (im, real), ns in zip(items, noises)]
        for d, e in zip(details, exp):
            assert e["top2_logit_margin"] > 2e-5, "seed gives a near-tie in the oracle's own logits (%.2e): pick another" % e["top2_logit_margin"]
            assert np.array_equal(d["tokens"], e["tokens"]), "sampled tokens differ from the oracle"
            assert float(np.abs(d["scores"] - e["scores"]).max()) <= 1e-4 + 1e-4 * float(np.abs(e["scores"]).max())
            assert (d["r50"], d["r100"]) == (e["r50"], e["r100"]), (literal, d["r50"], d["r100"], e["r50"], e["r100"])
        assert r50 == float(np.mean([e["r50"] for e in exp])) and r100 == float(np.mean([e["r100"] for e in exp]))
        lines = open(str(tmp_path / "recalls.txt")).read().splitlines()
        assert float(lines[0]) == r50 and float(lines[1]) == r100 and lines[2].startswith("# ordering:")
        assert ("reference_literal" in lines[2]) == literal
        print("literal=%s: R@50 %.4f  R@100 %.4f (oracle equal); distinct sampled triples per image: %s" %
              (literal, r50, r100, [len(set(map(tuple, d["tokens"]))) for d in details]))
    # the intended ordering finds more than the literal one here (the literal one only ever looks at sample 0)
    assert exp is not None


def test_validation_loss_early_stop_and_test_at_end(tmp_path):
    """train(): the critic's cost on a validation batch every `validate_every` iterations == the oracle's d_loss on the same rows;
    `patience` consecutive increases end the loop (train.py:375-384); then the evaluation runs (train.py:387-388)."""
    B, S, V = 8, 64, 50
    gan = _gan(tmp_path, B, S, V)
    images0, _ = gan._next_batch(0)
    gan._constructOps(images0)
    # (1) the validation loss itself against the oracle, on the weights as they stand
    gp, dp = gan.g.state_dict(full_names=False), gan.d.state_dict(full_names=False)
    vgen = torch.Generator().manual_seed(11)
    got = gan.validation_loss(0, vgen)
    images, labels = gan._val_batch(0)
    VB = gan._val_rows()
    assert VB == B // 2 and tuple(images.shape) == (VB, S, S, 3)         # VAL_BATCH_SIZE rows through the same variables (train.py:29-30)
    assert gan.val_step.B == VB and gan.val_step.G.arena is gan.step.G.arena and gan.val_step.D.arena is gan.step.D.arena
    g2 = torch.Generator().manual_seed(11)
    noise, alpha = torch.randn((VB, 512), generator=g2), torch.rand((VB,), generator=g2)
    onehot = torch.nn.functional.one_hot(labels[:VB].cpu(), V).float()
    cost, _ = O.d_loss(gp, dp, images[:VB].cpu(), onehot, noise, alpha.reshape(VB, 1, 1), 10.0)
    assert abs(got - float(cost)) <= 1e-4 + 1e-4 * abs(float(cost)), (got, float(cost))
    # (2) the stop rule inside train(): feed a rising validation loss
    seq = iter([5.0, 4.0, 4.5, 4.6, 4.7, 1.0, 1.0])
    gan.validation_loss = lambda k, gen: next(seq)
    tested = []
    gan.test = lambda *a, **k: tested.append(1) or (0.0, 0.0)
    gan.train(max_iterations=40, log_every=1000, validate_every=2, test_at_end=True)
    # validations after iterations 0, 2, 4, 6, 8 (0-based): 5, 4, 4.5 (1), 4.6 (2), 4.7 (3) -> stop after the 9th iteration
    assert gan.stopped_early and gan.itr == 9 and [v for _, v in gan.val_history] == [5.0, 4.0, 4.5, 4.6, 4.7]
    assert tested == [1] and os.path.exists(gan._ckpt_path())
    log = [json.loads(l) for l in open(os.path.join(str(tmp_path / "logs"), "losses.jsonl"))]
    assert [r["val_disc_loss"] for r in log if "val_disc_loss" in r] == [5.0, 4.0, 4.5, 4.6, 4.7]
