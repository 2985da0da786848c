"""-m gpu: the drop-in boundary (architectures.* classes with the reference's signatures, train.py plumbing)
against the oracle on BASELINE.json configs[0] (8 x 64x64 images, vocab 50).  Tolerance 1e-4 + 1e-4*|ref| (fp32)."""
import os

import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu


def close(a, b, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    tol = 1e-4 + 1e-4 * float(b.abs().max())
    assert float((a - b).abs().max()) <= tol, "%s: %.3e > %.3e" % (what, float((a - b).abs().max()), tol)


def test_generator_discriminator_classes_match_oracle():
    from architectures.generator_with_attention import Generator
    from architectures.discriminator_with_attention import Discriminator
    B, S, V = 8, 64, 50
    gp, dp = O.init_params("G", V, S), O.init_params("D", V, S)      # same initialisers + seed as the product default
    images, labels, onehot = O.synth_batch(B, S, V)
    noise = O.synth_noise(B, 0)
    g = Generator(V)
    logits = g.build_generator(images.cuda(), True, noise=noise.cuda())
    ref = O.generator_forward(gp, images, noise)
    close(logits, ref, "Generator.build_generator")
    assert tuple(g.downsampled.shape) == (B, 4, 4, 512) and tuple(g.flattened_context.shape) == (B, 16 * 512)
    assert tuple(g.partially_flattened_context.shape) == (B, 16, 512) and tuple(g.alpha.shape) == (B, 16)
    # weights are shared across builds (AUTO_REUSE): a second build gives the same result
    close(g.build_generator(images.cuda(), True, noise=noise.cuda()), ref, "second build")
    # attentionMechanism on an arbitrary state
    feat = O.encoder(gp, images)
    ctx_flat, ctx, m = O.context_views(feat)
    c = torch.randn(B, 512, generator=torch.Generator().manual_seed(5))
    z_ref, al_ref = O.attention(gp, ctx_flat, ctx, c)
    z = g.attentionMechanism((c.cuda(), c.cuda()))
    close(z, z_ref, "Generator.attentionMechanism")
    close(g.alpha, al_ref, "alpha")
    # critic
    d = Discriminator(V, dp["W"].clone())
    out = d.build_discriminator(onehot.cuda(), images.cuda())
    close(out, O.discriminator_forward(dp, onehot, images), "Discriminator.build_discriminator(real)")
    out = d.build_discriminator(ref.cuda(), images.cuda())
    close(out, O.discriminator_forward(dp, ref, images), "Discriminator.build_discriminator(fake)")
    assert tuple(out.shape) == (B, 3, 1)
    assert d.embedding_matrix.data_ptr() == d.net.arena.views["W"].data_ptr()
    with pytest.raises(RuntimeError):
        Generator(V).build_generator(images, True)          # CPU tensor: no fallback


def test_model_objects_are_batch_dynamic_on_one_set_of_weights():
    """The reference graph is batch-dynamic (generator_with_attention.py:74-75: reshape(..., [-1, ...])) and train.py feeds it B,
    B / 2 (validation, :29-30, 199-203) and B / 2 x 8 (test, :297-298) rows through the same variables.  One Generator / Discriminator:
    B = 8 and B = 4 against the oracle (logits 1e-4, tokens exact); then a critic + generator update at B = 8 (Adam on the shared
    arenas) must be what the B = 4 build sees - again against the oracle with the UPDATED weights."""
    from architectures.generator_with_attention import Generator
    from architectures.discriminator_with_attention import Discriminator
    from sgg_amd.step import GanStep
    S, V = 64, 50
    gp, dp = O.init_params("G", V, S), O.init_params("D", V, S)
    g, d = Generator(V), Discriminator(V, dp["W"].clone())
    toks = torch.empty((8, 3), dtype=torch.int64, device="cuda")

    def check(B, gp, dp, what):
        images, labels, onehot = O.synth_batch(B, S, V)
        noise = O.synth_noise(B, 3)
        logits = g.build_generator(images.cuda(), True, noise=noise.cuda())
        ref = O.generator_forward(gp, images, noise)
        close(logits, ref, "%s: Generator at B = %d" % (what, B))
        assert O.top2_margin(ref) > 2e-5, "near-tie in the oracle's own logits: pick another seed"
        g.net.K.argmax_rows(logits, toks[:B].view(-1))
        assert torch.equal(toks[:B].cpu(), O.argmax_tokens(ref)), "%s: tokens at B = %d" % (what, B)
        assert tuple(g.downsampled.shape) == (B, 4, 4, 512) and tuple(g.alpha.shape) == (B, 16)
        out = d.build_discriminator(ref.cuda(), images.cuda())
        close(out, O.discriminator_forward(dp, ref, images), "%s: Discriminator at B = %d" % (what, B))
        out = d.build_discriminator(onehot.cuda(), images.cuda())
        close(out, O.discriminator_forward(dp, onehot, images), "%s: Discriminator(real) at B = %d" % (what, B))
        return images, labels

    images8, labels8 = check(8, gp, dp, "initial weights")
    check(4, gp, dp, "initial weights")
    check(8, gp, dp, "initial weights, back at 8")          # the buffers of the first batch size are intact
    assert set(g._nets) == {8, 4} and g._nets[4].arena is g.net.arena and g._nets[4].grad_flat is g.net.grad_flat
    assert d._nets[4].m_flat is d.net.m_flat and d.embedding_matrix.data_ptr() == d.net.arena.views["W"].data_ptr()
    # one critic update and one generator update at B = 8 on the very same objects
    gs = GanStep(g.net.K, V, S, 8, lam=10.0, G=g._ensure(images8.cuda()), D=d._ensure(images8.cuda()))
    gs.critic_step(images8.cuda(), labels8.cuda(), O.synth_noise(8, 0).cuda(), O.synth_alpha(8, 0).reshape(8).cuda())
    gs.generator_step(images8.cuda(), O.synth_noise(8, 1).cuda())
    gs.flush()
    assert g.net.adam_t == 1 and d.net.adam_t == 1 and g._nets[4].adam_t == 1
    gp2, dp2 = g.state_dict(full_names=False), d.state_dict(full_names=False)
    assert float((gp2["conv2d/kernel"] - gp["conv2d/kernel"]).abs().max()) > 0 and float((dp2["W"] - dp["W"]).abs().max()) > 0
    check(4, gp2, dp2, "after the update at B = 8")          # the B = 4 encoder re-derives its operand formats from the new weights
    check(8, gp2, dp2, "after the update at B = 8")
    # the per-batch-size buffer cache is bounded (owner + the most recently used others): ever-changing batch sizes do not grow it
    for b in (2, 3, 5, 6, 7):
        g.build_generator(torch.zeros((b, S, S, 3), device="cuda"))
    assert len(g._nets) <= g.max_cached_batch_sizes and g._nets[8] is g.net and 7 in g._nets and 4 not in g._nets
    check(4, gp2, dp2, "B = 4 re-created after its eviction")
    with pytest.raises(ValueError):
        g.build_generator(torch.zeros((2, 96, 96, 3), device="cuda"))       # the spatial size is static, as in the reference (:15)


def test_train_entry_point_synthetic(tmp_path):
    import train as T
    gan = T.SceneGraphGAN(str(tmp_path / "ck"), str(tmp_path / "logs"), None, None, None, None, None,
                          critic_iters=2, batch_size=8, lambda_=10, resume=False, synthetic=(8, 64, 50))
    gan.train(max_iterations=2, log_every=1)
    assert os.path.exists(gan._ckpt_path())
    d = gan.step.d_losses.cpu()
    assert torch.isfinite(d).all() and gan.step.D.adam_t == 4 and gan.step.G.adam_t == 2
    w_before = gan.g.net.arena.flat.clone()
    images, _ = gan._next_batch(0)
    toks, words = gan.sample_triples(images)
    assert tuple(toks.shape) == (8, 3) and len(words) == 8 and len(words[0]) == 3
    # resume picks the weights and Adam state up again
    gan2 = T.SceneGraphGAN(str(tmp_path / "ck"), str(tmp_path / "logs"), None, None, None, None, None,
                           critic_iters=2, batch_size=8, lambda_=10, resume=True, synthetic=(8, 64, 50))
    gan2.train(max_iterations=2)
    assert gan2.itr == 2 and torch.equal(gan2.g.net.arena.flat, w_before) and gan2.step.D.adam_t == 4


def test_real_data_pipeline_and_evaluation(tmp_path):
    """f1 + f3 rows: the reference's on-disk formats (ims_to_triples.json, vocab.json, word_embeddings.npy [V,300] float64,
    image_means/stds.txt), JPEG -> 221x221 -> standardise, one training iteration, checkpoint, R@50/R@100."""
    import json
    import numpy as np
    from PIL import Image
    import train as T
    rng = np.random.RandomState(0)
    V = 20
    ims = {}
    for i in range(12):
        path = str(tmp_path / ("im%d.jpg" % i))
        Image.fromarray(rng.randint(0, 255, (40 + i, 50, 3), dtype=np.uint8)).save(path)
        ims[path] = rng.randint(0, V, (3, 3)).tolist()
    (tmp_path / "ims.json").write_text(json.dumps(ims))
    (tmp_path / "vocab.json").write_text(json.dumps({"w%d" % i: i for i in range(V)}))
    np.save(str(tmp_path / "emb.npy"), rng.uniform(-0.1, 0.1, (V, 300)))
    (tmp_path / "means.txt").write_text("119.6\n115.1\n106.1\n")
    (tmp_path / "stds.txt").write_text("30.4\n30.5\n36.7\n")
    gan = T.SceneGraphGAN(str(tmp_path / "ck"), str(tmp_path / "logs"), str(tmp_path / "ims.json"), str(tmp_path / "vocab.json"),
                          str(tmp_path / "emb.npy"), str(tmp_path / "means.txt"), str(tmp_path / "stds.txt"),
                          critic_iters=1, batch_size=4, lambda_=10, resume=False)
    images, labels = gan._next_batch(0)
    assert tuple(images.shape) == (4, 221, 221, 3) and tuple(labels.shape) == (4, 3)
    assert abs(float(images.mean())) < 3.0            # standardised
    # the prefetching loader (decode threads + pinned double buffer + copy stream) delivers the same batches
    loader = gan._prefetcher(0, 2, workers=2)
    im_a, lab_a = next(loader)
    im_b, lab_b = next(loader)
    # (the device-side resize kernel restates the host arithmetic operation by operation: equal to a few ulps)
    print("loader vs synchronous pipeline: max |d| = %.3e" % float((im_a - images).abs().max()))
    assert torch.equal(lab_a, labels) and float((im_a - images).abs().max()) <= 5e-5
    images1, labels1 = gan._next_batch(1)
    assert torch.equal(lab_b, labels1) and float((im_b - images1).abs().max()) <= 5e-5
    with pytest.raises(StopIteration):
        next(loader)
    gan.train(max_iterations=1)
    assert torch.isfinite(gan.step.d_losses).all() and os.path.exists(gan._ckpt_path())
    # evaluation on the test split of the real-data files, against the oracle's restatement of train.py:297-335 on the same
    # weights, decoded images and noise (tests/test_f1_f3_gpu.py does the same at 128 samples per image)
    from oracle import eval_ref as ER
    (r50, r100), details = gan.test(max_images=1, out_path=str(tmp_path / "recalls.txt"), return_details=True)
    key, triples = gan.test_items[0]
    gen = torch.Generator().manual_seed(gan.seed + 123)
    n_samples = gan.TEST_BATCH_MULTIPLIER * gan.TEST_BATCH_SIZE
    noises = [torch.randn((4, 512), generator=gen) for _ in range(-(-n_samples // 4))]
    exp = ER.evaluate_image(gan.g.state_dict(full_names=False), gan.d.state_dict(full_names=False), gan._parseFunction(key), triples, noises)
    assert np.array_equal(details[0]["tokens"], exp["tokens"][:n_samples])
    assert float(np.abs(details[0]["scores"] - exp["scores"][:n_samples]).max()) <= 1e-4 + 1e-4 * float(np.abs(exp["scores"]).max())
    assert (r50, r100) == (exp["r50"], exp["r100"]) and os.path.exists(str(tmp_path / "recalls.txt"))
