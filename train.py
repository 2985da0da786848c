"""Entry point mirroring the reference's train.py (class SceneGraphGAN + the same CLI flags), MI355X-native.

    python train.py --synthetic 64,224,1000 --max_iterations 20           # no Visual Genome files needed
    python train.py --path_to_ims_to_triples ... --path_to_vocab ... --path_to_word_embeddings ...

Reference: train.py:17-422.  Kept: constructor signature (:23-24), `_Generator` / `_Discriminator` wrappers with
shared weights (:85-93), the loss / optimiser definition (:239-266) and the loop body (:362-368: CRITIC_ITERS critic
updates then one generator update on the same minibatch, fresh noise / alpha per update).  The TensorFlow graph /
tfgan / tf.data machinery is replaced by sgg_amd.step.GanStep (hand-written HIP kernels behind libsgg_hip.so).
Deviations, all documented in SURVEY.md Appendix C: flags are passed by keyword (the reference swaps batch_size and
critic_iters, C-2); the checkpoint directory is not wiped when resuming and checkpoints are actually written (C-5);
evaluation uses the trained weights (C-4).  Multi-GPU: launch with torch.distributed.run, one process per GPU.
"""
import os, sys
sys.path.append(os.getcwd())

import argparse
import json
import time

import numpy as np
import torch

from architectures.generator_with_attention import Generator
from architectures.discriminator_with_attention import Discriminator

import sgg_amd  # noqa: F401
from sgg_amd import dp as dpmod
from sgg_amd.api import kernels_for
from sgg_amd.data import PrefetchLoader, ShuffledStream, parse_image
from sgg_amd.params import EMBED_DIM
from sgg_amd.step import GanStep


class ValidationEarlyStop(object):
    """The convergence test of train.py:358-384: `last_loss = inf; convergence_count = 0`, then per validation
    `if last_loss < loss: count += 1 else: count = 0; if count == 3: break; last_loss = loss`."""

    def __init__(self, patience=3):
        self.patience, self.count, self.last = int(patience), 0, float("inf")

    def update(self, loss):
        """True = stop training now."""
        self.count = self.count + 1 if self.last < loss else 0
        if self.count == self.patience:
            return True
        self.last = loss
        return False


class SceneGraphGAN(object):

    ############################################################
    ## All init methods
    ############################################################
    def __init__(self, checkpoints_dir, summaries_dir, path_to_ims_to_triples, path_to_vocab, path_to_word_embeddings,
                 path_to_image_means, path_to_image_stds, critic_iters, batch_size, lambda_, resume,
                 synthetic=None, device=None, seed=0, two_streams=True, reuse_g_encoder=True, shuffle_buffer=True):
        # Hyperparameters (train.py:26-32)
        self.CRITIC_ITERS = int(critic_iters)
        self.BATCH_SIZE = int(batch_size)
        self.VAL_BATCH_SIZE = self.BATCH_SIZE // 2
        self.TEST_BATCH_SIZE = self.BATCH_SIZE // 2
        self.TEST_BATCH_MULTIPLIER = 8
        self.LAMBDA = float(lambda_)
        self.resume = bool(resume)
        # two-stream schedule of step.GanStep (D's encoder beside G's forward, filter gradients beside the dgrad -> LayerNorm
        # chain): same kernels, bit-identical results (tests/test_concurrency_gpu.py), +4 % triples/s
        self.two_streams = bool(two_streams)
        self.reuse_g_encoder = bool(reuse_g_encoder)
        self.shuffle_buffer = bool(shuffle_buffer)      # tf.data shuffle(buffer_size = 10 * batch) on the repeated stream (train.py:176-179)
        self.checkpoints_dir, self.summaries_dir = checkpoints_dir, summaries_dir
        self.rank, self.world, local = dpmod.init_from_env()
        self.device = torch.device(device if device is not None else "cuda:%d" % local)
        torch.cuda.set_device(self.device)
        if self.rank == 0:
            os.makedirs(checkpoints_dir, exist_ok=True)
            os.makedirs(summaries_dir, exist_ok=True)
        self.seed = seed
        if synthetic is not None:
            B, S, V = synthetic
            self.BATCH_SIZE, self.image_size = B, S
            self.vocab = {"w%d" % i: i for i in range(V)}
            g = torch.Generator().manual_seed(3)
            self.embeddings = (torch.rand((V, EMBED_DIM), generator=g) * 0.2 - 0.1).numpy()   # map_files_to_triples.py:24
            self.dataset = None
        else:
            self.image_size = 221                                           # train.py:171
            with open(path_to_ims_to_triples, "r") as f:
                self.ims_to_triples = json.load(f)
            with open(path_to_vocab, "r") as f:
                self.vocab = json.load(f)
            self.embeddings = np.load(path_to_word_embeddings)
            self._loadImageMeans(path_to_image_means, path_to_image_stds)
            self.dataset = self._gatherFiles()
        self._createStringMappings()
        self.g = Generator(len(self.vocab))
        self.d = Discriminator(len(self.vocab), torch.as_tensor(self.embeddings, dtype=torch.float32))
        self.step = None
        self.val_step = None
        self.itr = 0

    def _createStringMappings(self):
        self.reverse_vocab = {y: x for x, y in self.vocab.items()}                          # train.py:76-80

    def _loadImageMeans(self, path_means, path_stds):
        with open(path_means) as f:
            self.image_means = torch.tensor([float(l.strip()) for l in f if l.strip()])
        with open(path_stds) as f:
            self.image_stds = torch.tensor([float(l.strip()) for l in f if l.strip()])

    def _Generator(self, images, is_training=True):
        return self.g.build_generator(images, is_training)

    def _Discriminator(self, triple_input, images, is_training=True):
        return self.d.build_discriminator(triple_input, images, is_training)

    ############################################################
    ## Data (train.py:114-226); synthetic mode needs no files
    ############################################################
    def _gatherFiles(self):
        keys = list(self.ims_to_triples.keys())
        train_keys = keys[:int(0.9 * len(keys))]
        self.test_items = [(k, self.ims_to_triples[k]) for k in keys[int(0.9 * len(keys)):] if len(self.ims_to_triples[k])]
        files, labels = [], []
        for k in train_keys:
            for t in self.ims_to_triples[k]:
                files.append(k)
                labels.append(t)
        perm = np.random.RandomState(self.seed).permutation(len(files))
        files, labels = [files[i] for i in perm], np.asarray(labels, dtype=np.int64)[perm]
        thr = int(0.88 * len(files))
        self.max_iterations = 5 * thr                                       # train.py:159
        self.write_iterations = 10                                          # train.py:161
        self.validate_iterations = max(1, int(thr / 50))                    # train.py:162
        return {"train": (files[:thr], labels[:thr]), "val": (files[thr:], labels[thr:])}

    def _streams(self):
        """The example order of the reference's datasets (train.py:176-179): the shuffled list repeated for ever, through a rolling
        shuffle buffer of 10 batches (sgg_amd.data.ShuffledStream).  One stream per consumer (the prefetching loader's producer
        thread, the synchronous path, validation): same seed -> same order."""
        if getattr(self, "_stream_cache", None) is None:
            B, VB = self.BATCH_SIZE, self._val_rows()
            n_tr = len(self.dataset["train"][0]) if self.dataset is not None else 0
            n_va = len(self.dataset["val"][0]) if self.dataset is not None else 0
            mk = lambda n, b, salt: ShuffledStream(n, 10 * b * self.world, seed=self.seed + salt) if (n and self.shuffle_buffer) else None
            self._stream_cache = {"loader": mk(n_tr, B, 11), "sync": mk(n_tr, B, 11), "val": mk(n_va, VB, 12)}
        return self._stream_cache

    def _val_rows(self):
        """VAL_BATCH_SIZE = BATCH_SIZE / 2 (train.py:30, Python-2 integer division); the model objects take any batch size on one
        set of weights (sgg_amd/api.py), so the validation batch runs at its own size, as in the reference graph."""
        return max(1, self.BATCH_SIZE // 2)

    def _parseFunction(self, filename):
        """JPEG decode -> tf.image.resize_images([221, 221]) (TF-1.x bilinear, align_corners=False, no antialiasing) ->
        (x - mean) / std (train.py:167-172), on the host: sgg_amd/data.py."""
        return torch.from_numpy(parse_image(filename, self.image_means.numpy(), self.image_stds.numpy()))

    def _batch_indices(self, it, stream="sync"):
        """Example indices of iteration `it` on this rank (rank r takes rows [r*B, (r+1)*B) of the global batch): batch `it` of the
        repeated + buffer-shuffled stream (train.py:176-179), or of the plain cyclic walk with shuffle_buffer=False."""
        B, n = self.BATCH_SIZE, len(self.dataset["train"][0])
        st = self._streams()[stream]
        if st is not None:
            return st.batch(it, B, self.rank, self.world)
        return [(it * B * self.world + self.rank * B + j) % n for j in range(B)]

    def _next_batch(self, it):
        B = self.BATCH_SIZE
        if self.dataset is None:
            g = torch.Generator().manual_seed(self.seed + 17 * it + 1000 * self.rank)
            images = torch.randn((B, self.image_size, self.image_size, 3), generator=g)
            labels = torch.randint(0, len(self.vocab), (B, 3), generator=g, dtype=torch.int64)
        else:
            files, labs = self.dataset["train"]
            idx = self._batch_indices(it)
            images = torch.stack([self._parseFunction(files[i]) for i in idx])
            labels = torch.from_numpy(labs[idx])
        return images.to(self.device), labels.to(self.device)

    def _val_batch(self, k):
        """Validation batch k (the live validation iterator of train.py:199-203): VAL_BATCH_SIZE examples of the validation split in
        its own repeated + shuffled order."""
        VB = self._val_rows()
        if self.dataset is None:
            g = torch.Generator().manual_seed(self.seed + 5000 + 17 * k + 1000 * self.rank)
            images = torch.randn((VB, self.image_size, self.image_size, 3), generator=g)
            labels = torch.randint(0, len(self.vocab), (VB, 3), generator=g, dtype=torch.int64)
        else:
            files, labs = self.dataset["val"]
            st = self._streams()["val"]
            idx = st.batch(k, VB, self.rank, self.world) if st is not None else \
                [(k * VB * self.world + self.rank * VB + j) % len(files) for j in range(VB)]
            images = torch.stack([self._parseFunction(files[i]) for i in idx])
            labels = torch.from_numpy(labs[idx])
        return images.to(self.device), labels.to(self.device)

    def validation_loss(self, k, gen):
        """np.mean(sess.run(self.disc_cost, feed_dict = {handle: val_handle})) (train.py:377): the critic's cost on validation batch k with
        fresh noise / alpha, no update; averaged over the data-parallel ranks so that every rank takes the same early-stop decision."""
        images, labels = self._val_batch(k)
        VB = self._val_rows()
        noise = torch.randn((VB, 512), generator=gen).to(self.device)
        alpha = torch.rand((VB,), generator=gen).to(self.device)
        if self.val_step is None:
            # the same variables at the validation batch size (train.py:199-203 feeds the validation iterator through the same graph)
            self.val_step = GanStep(kernels_for(self.device), len(self.vocab), self.image_size, VB, lam=self.LAMBDA,
                                    G=self.g._ensure(images), D=self.d._ensure(images))
        self.step.flush()           # (a deferred optimiser step of the training networks changes the weights validation reads)
        loss = self.val_step.critic_loss(images, labels, noise, alpha)[0:1].clone()
        if self.world > 1:
            torch.distributed.all_reduce(loss)
            loss /= self.world
        return float(loss.item())

    def _prefetcher(self, start, stop, workers=16):
        """tf.contrib.data.map_and_batch + prefetch (train.py:181-187) as decode threads + a pinned double buffer whose
        host-to-device copy runs on its own stream while the previous batch trains."""
        files, labs = self.dataset["train"]
        return PrefetchLoader(files, labs, self.BATCH_SIZE, lambda it: self._batch_indices(it, "loader"), self.image_means.numpy(),
                              self.image_stds.numpy(), self.device, stop, start=start, workers=workers, processes=workers > 2)

    ############################################################
    ## Saving
    ############################################################
    def _ckpt_path(self):
        return os.path.join(self.checkpoints_dir, "model.ckpt.pt")

    def _saveModel(self):
        self.step.flush()
        if self.rank == 0:
            stopper = getattr(self, "_stopper", None)
            torch.save({"itr": self.itr, "G": self.g.state_dict(), "D": self.d.state_dict(),
                        "G_adam": (self.step.G.m_flat.cpu(), self.step.G.v_flat.cpu(), self.step.G.adam_t),
                        "D_adam": (self.step.D.m_flat.cpu(), self.step.D.v_flat.cpu(), self.step.D.adam_t),
                        # the validation state of the loop (train.py:358-384): last loss, consecutive increases, batches consumed
                        "val": {"last": stopper.last if stopper else float("inf"), "count": stopper.count if stopper else 0,
                                "history": list(getattr(self, "val_history", []))}}, self._ckpt_path())

    def _loadModel(self):
        ck = torch.load(self._ckpt_path(), map_location="cpu")
        self.g.load_state_dict(ck["G"])
        self.d.load_state_dict(ck["D"])
        for net, key in ((self.step.G, "G_adam"), (self.step.D, "D_adam")):
            net.m_flat.copy_(ck[key][0]); net.v_flat.copy_(ck[key][1]); net.adam_t = ck[key][2]
        self.itr = ck["itr"]
        self._resumed_val = ck.get("val")

    ############################################################
    ## Training (train.py:341-388)
    ############################################################
    def _constructOps(self, images):
        """Build both networks for this static shape and wire the WGAN-GP step (train.py:231-266)."""
        B, S, V = images.shape[0], images.shape[1], len(self.vocab)
        g_net, d_net = self.g._ensure(images), self.d._ensure(images)
        reducer = dpmod.GradReducer() if self.world > 1 else None
        self.step = GanStep(kernels_for(self.device), V, S, B, lam=self.LAMBDA, G=g_net, D=d_net, reducer=reducer,
                            overlap_streams=self.two_streams)

    def train(self, max_iterations=None, log_every=10, save_every=0, validate_every=None, test_at_end=None, patience=3):
        """train.py:341-388.  Every `validate_every` iterations (default: the reference's len(train) / 50 on real data, off in
        synthetic mode) the critic's cost on a validation batch is compared with the previous one: `patience` (3) consecutive
        increases end the training (train.py:375-384); afterwards the model is evaluated (`print "Testing"; self.test(sess)`,
        train.py:387-388) - by default on real data only."""
        images, labels = self._next_batch(0)
        self._constructOps(images)
        if self.resume and os.path.exists(self._ckpt_path()):
            self._loadModel()
        n_it = max_iterations if max_iterations is not None else getattr(self, "max_iterations", 1000)
        if validate_every is None:
            validate_every = getattr(self, "validate_iterations", 0) if self.dataset is not None else 0
        if test_at_end is None:
            test_at_end = self.dataset is not None
        gen = torch.Generator().manual_seed(self.seed + 7 + self.rank)
        vgen = torch.Generator().manual_seed(self.seed + 70007 + self.rank)
        log = open(os.path.join(self.summaries_dir, "losses.jsonl"), "a") if self.rank == 0 else None
        B, t0, itr0 = self.BATCH_SIZE, time.time(), self.itr
        loader = self._prefetcher(self.itr, n_it) if self.dataset is not None else None
        stopper, self.stopped_early, self.val_history = ValidationEarlyStop(patience), False, []
        rv = getattr(self, "_resumed_val", None)
        if rv is not None:          # a resumed run carries on with the validation iterator and the early-stop counters where it stopped
            stopper.last, stopper.count, self.val_history = rv["last"], rv["count"], [tuple(x) for x in rv["history"]]
            self._resumed_val = None
        self._stopper = stopper
        try:
            while self.itr < n_it:
                images, labels = next(loader) if loader is not None else self._next_batch(self.itr)
                # every update of an iteration sees the same minibatch (train.py:175-190) and G's weights change only at its end: G's
                # encoder runs once per iteration (exact; 10 of 11 encoder forwards of G saved at CRITIC_ITERS = 10)
                with self.step.iteration(reuse_g_encoder=self.reuse_g_encoder):
                    for _ in range(self.CRITIC_ITERS):                                  # train.py:364-365
                        noise = torch.randn((B, 512), generator=gen).to(self.device)
                        alpha = torch.rand((B,), generator=gen).to(self.device)
                        self.step.critic_step(images, labels, noise, alpha)
                    noise = torch.randn((B, 512), generator=gen).to(self.device)
                    self.step.generator_step(images, noise)                             # train.py:368
                itr = self.itr                                                          # the reference's 0-based loop variable
                self.itr += 1
                if log is not None and self.itr % log_every == 0:
                    d, g = self.step.d_losses.cpu().tolist(), self.step.g_losses.cpu().tolist()
                    rate = B * self.world * (self.itr - itr0) / (time.time() - t0)      # of this run (a resumed run starts at itr0 > 0)
                    rec = {"itr": self.itr, "disc_loss": d[0], "gen_loss": -g[3], "gp": d[2], "triples_per_s": rate}
                    log.write(json.dumps(rec) + "\n"); log.flush()
                    print(rec)
                if save_every and self.itr % save_every == 0:
                    self._saveModel()
                if validate_every and itr % validate_every == 0:                        # train.py:375-384
                    loss = self.validation_loss(len(self.val_history), vgen)
                    self.val_history.append((itr, loss))
                    if log is not None:
                        log.write(json.dumps({"itr": self.itr, "val_disc_loss": loss}) + "\n"); log.flush()
                    if stopper.update(loss):
                        self.stopped_early = True
                        break
        finally:
            if loader is not None:
                loader.close()
        self._saveModel()
        if test_at_end:
            print("Testing")                                                            # train.py:387-388
            return self.test()

    ############################################################
    ## Testing (train.py:294-335)
    ############################################################
    def _recall(self, fake, real, N):
        return float(len(set(map(tuple, fake)).intersection(set(map(tuple, real))))) / N

    @staticmethod
    def _rank(scores, reference_literal=False):
        """Order of the sampled triples for R@k (train.py:321-323).  Intended semantics (default): ascending mean critic
        score over the three steps.  reference_literal=True reproduces what the reference's code does: its score array has
        shape [N, 1] (np.mean(disc_scores, axis=1) of [B, 3, 1], train.py:315), so `argsort()` sorts the length-1 last axis
        and returns zeros - every "top-k" entry is sample 0 (reference quirk C-11)."""
        scores = np.asarray(scores, dtype=np.float64).reshape(-1)
        if reference_literal:
            return np.zeros(len(scores), dtype=np.int64)
        return np.argsort(scores, kind="stable")

    def recalls(self, fake, scores, real, reference_literal=False):
        """(R@50, R@100) of one image: fake [N,3] sampled token triples, scores [N] mean critic outputs, real [M,3] true
        triples.  Set semantics of train.py:294-295: duplicates collapse, the denominators are the constants 50 and 100."""
        order = self._rank(scores, reference_literal)
        fake, real = np.asarray(fake), np.asarray(real, dtype=np.int64).reshape(-1, 3)
        return self._recall(fake[order[:50]], real, 50.0), self._recall(fake[order[:100]], real, 100.0)

    def test(self, max_images=None, out_path="recalls.txt", reference_literal=False, items=None, return_details=False):
        """R@50 / R@100 (train.py:297-335): per test image TEST_BATCH_MULTIPLIER x TEST_BATCH_SIZE generator samples, each scored by
        the mean critic output over the three steps (`np.mean(disc_scores, axis=1)`, :315), ordered by score, the first 50 / 100
        compared as sets with the image's true triples (:294-295, :325-326), averaged over the images, written to recalls.txt.

        ORDERING.  Default (reference_literal=False): ascending mean critic score - what `score_accumulator.argsort()` (:321) is
        written to mean.  This is NOT what the reference's code computes: its score array has shape [N, 1], so argsort sorts the
        length-1 axis and every selected index is 0 (sample 0 repeated).  reference_literal=True reproduces
        that literally.  The mode in force is written into recalls.txt (third line) and returned with the details.
        Uses the trained weights (the reference's test ops use an untrained copy).
        items: optional list of (image [S,S,3] float tensor, true triples [[s,p,o], ...]) replacing the test split.
        return_details: also return, per image, the sampled tokens [N,3], their scores [N] and the two recalls."""
        if self.step is None:
            images, _ = self._next_batch(0)
            self._constructOps(images)
        self.step.flush()
        K = kernels_for(self.device)
        # TEST_BATCH_MULTIPLIER runs of TEST_BATCH_SIZE copies of the image (train.py:139-150, 311-318), at that batch size
        B = max(1, self.TEST_BATCH_SIZE)
        passes = self.TEST_BATCH_MULTIPLIER
        n_samples = passes * B
        if items is not None:
            items = list(items)[:max_images]
        elif self.dataset is None:
            g = torch.Generator().manual_seed(self.seed + 99)
            items = [(torch.randn((self.image_size, self.image_size, 3), generator=g),
                      torch.randint(0, len(self.vocab), (5, 3), generator=g).tolist()) for _ in range(max_images or 2)]
        else:
            items = [(self._parseFunction(k), t) for k, t in self.test_items[:max_images]]
        gen = torch.Generator().manual_seed(self.seed + 123)
        toks = torch.empty((B, 3), dtype=torch.int64, device=self.device)
        r50, r100, details = [], [], []
        for image, triples in items:
            images = image.unsqueeze(0).expand(B, -1, -1, -1).contiguous().to(self.device)
            fakes, scores = [], []
            for _ in range(passes):
                noise = torch.randn((B, 512), generator=gen).to(self.device)
                logits = self.g.build_generator(images, False, noise)
                K.argmax_rows(logits, toks.view(-1))
                d = self.d.build_discriminator(logits, images, False)
                fakes.append(toks.cpu().numpy().copy())
                scores.append(d.mean(dim=1).reshape(-1).cpu().numpy())
            fake, score = np.concatenate(fakes)[:n_samples], np.concatenate(scores)[:n_samples]
            a, b = self.recalls(fake, score, triples, reference_literal)
            r50.append(a)
            r100.append(b)
            details.append({"tokens": fake, "scores": score, "r50": a, "r100": b})
        res = (float(np.mean(r50)), float(np.mean(r100)))
        ordering = "reference_literal ([N,1] argsort: sample 0 repeated)" if reference_literal else "ascending mean critic score"
        if self.rank == 0 and out_path:
            with open(out_path, "w") as f:
                f.write("{}\n{}\n# ordering: {}\n".format(res[0], res[1], ordering))
        if self.rank == 0:
            print({"R@50": res[0], "R@100": res[1], "ordering": ordering, "images": len(items), "samples_per_image": n_samples})
        return (res, details) if return_details else res

    def sample_triples(self, images, noise=None):
        """tf.argmax(fake_inputs, -1) -> words (train.py:269-275), with the trained weights."""
        logits = self._Generator(images, False) if noise is None else self.g.build_generator(images, False, noise)
        toks = torch.empty((images.shape[0], 3), dtype=torch.int64, device=images.device)
        kernels_for(images.device).argmax_rows(logits, toks.view(-1))
        return toks, [[self.reverse_vocab.get(int(i), "UNK") for i in row] for row in toks.cpu()]


def _str2bool(v):
    """--resume of the reference is `type=bool` (train.py:410), which makes every non-empty string - "False" included - true."""
    if isinstance(v, bool):
        return v
    if str(v).strip().lower() in ("1", "true", "t", "yes", "y", "on"):
        return True
    if str(v).strip().lower() in ("0", "false", "f", "no", "n", "off", ""):
        return False
    raise argparse.ArgumentTypeError("expected a boolean, got %r" % (v,))


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoints_dir", help="Where to save the checkpoints", default="./checkpoints")
    parser.add_argument("--summaries_dir", help="Where to write the logs", default="./logs")
    parser.add_argument("--path_to_ims_to_triples", default="./dataset_creation/ims_to_triples.json")
    parser.add_argument("--path_to_vocab", default="./dataset_creation/vocab.json")
    parser.add_argument("--path_to_word_embeddings", default="./dataset_creation/word_embeddings.npy")
    parser.add_argument("--path_to_image_means", default="./dataset_creation/image_means.txt")
    parser.add_argument("--path_to_image_stds", default="./dataset_creation/image_stds.txt")
    parser.add_argument("--batch_size", default=64, help="Batch size defaults", type=int)
    parser.add_argument("--critic_iters", default=10, help="Number of critic iterations per generator iteration", type=int)
    parser.add_argument("--lambda", default=10, help="WGAN Lipschitz Penalty", type=float)
    parser.add_argument("--resume", default=False, nargs="?", const=True, type=_str2bool,
                        help="Resume from the last checkpoint (--resume, --resume True, --resume False)")
    parser.add_argument("--GPU", default="0", help="Which GPU to use (single-process runs)")
    parser.add_argument("--synthetic", default=None, help="B,S,V: train on synthetic tensors of that shape (no dataset files)")
    parser.add_argument("--max_iterations", default=None, type=int)
    parser.add_argument("--single_stream", action="store_true", help="serial launch order (default: two HIP streams)")
    parser.add_argument("--no_shuffle_buffer", action="store_true", help="walk the shuffled example list in order (default: through the "
                                                                         "reference's rolling shuffle buffer of 10 batches, train.py:178)")
    parser.add_argument("--validate_every", default=None, type=int, help="iterations between validation-loss checks (default: "
                                                                         "len(train) / 50 on real data as train.py:162, off with --synthetic)")
    parser.add_argument("--recompute_generator_encoder", action="store_true",
                        help="run G's encoder in every update like the reference graph (default: once per iteration, same result)")
    args = parser.parse_args()
    params = vars(args)

    if "LOCAL_RANK" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", "{}".format(params["GPU"]))
    synthetic = tuple(int(x) for x in params["synthetic"].split(",")) if params["synthetic"] else None
    gan = SceneGraphGAN(checkpoints_dir=params["checkpoints_dir"], summaries_dir=params["summaries_dir"],
                        path_to_ims_to_triples=params["path_to_ims_to_triples"], path_to_vocab=params["path_to_vocab"],
                        path_to_word_embeddings=params["path_to_word_embeddings"],
                        path_to_image_means=params["path_to_image_means"], path_to_image_stds=params["path_to_image_stds"],
                        critic_iters=params["critic_iters"], batch_size=params["batch_size"], lambda_=params["lambda"],
                        resume=params["resume"], synthetic=synthetic, two_streams=not params["single_stream"],
                        reuse_g_encoder=not params["recompute_generator_encoder"], shuffle_buffer=not params["no_shuffle_buffer"])
    gan.train(max_iterations=params["max_iterations"], validate_every=params["validate_every"])
