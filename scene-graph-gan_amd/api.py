"""Host-side mirror of the reference's model classes (shared by architectures/*.py): lazily builds the network for the
image size it first sees and for every batch size it is called with, shares ONE set of weights across all builds
(tf.AUTO_REUSE, train.py:86,91) and exposes the attributes the reference sets on `self`
(generator_with_attention.py:16,68,74,75)."""
from __future__ import annotations

import collections

import torch

from .lib import HipKernels
from .params import EMBED_DIM, FEAT_C, init_state_dict
from .step import Network

_KERNELS = {}


def kernels_for(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("scene-graph-gan_amd runs on MI355X GPUs only (tensor on %s): there is no CPU fallback" % device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _KERNELS:
        _KERNELS[key] = HipKernels("cuda:%d" % key)
    return _KERNELS[key]


class NetworkHandle(object):
    def __init__(self, kind, vocab_size):
        self.kind, self.vocab_size = kind, int(vocab_size)
        self.net = None              # the network of the first batch size seen (owner of the parameter arenas)
        # batch size -> Network sharing those arenas, least recently used first.  Every entry owns the encoder / head activation buffers
        # of its batch size (multi-GB at 224x224), so the cache is BOUNDED: the owner plus the max_cached_batch_sizes - 1 most recently
        # used others (train.py needs three: B, VAL_BATCH_SIZE = B / 2 and the test pass); a caller that feeds ever-changing batch sizes
        # (a last partial batch, ad-hoc sampling) re-creates buffers instead of growing until the allocator fails.  An evicted Network
        # that a caller still holds (a GanStep built on it) stays valid - it only leaves this cache.
        self._nets = collections.OrderedDict()
        self.max_cached_batch_sizes = 4
        self._last = None            # the network of the most recent build (attention / attributes refer to it)
        self.embedding_matrix = None
        self.init_seed = 3

    def _ensure(self, images):
        """The network for this batch size, created on first use.  The variables (parameters with the reference's initialisers,
        gradient and Adam slots) exist once; every batch size gets activation buffers of its own on top of them - the reference
        graph is batch-dynamic (generator_with_attention.py:74-75) and serves B, B/2 and B/2 x 8 rows (train.py:29-30, 199-203,
        297-298).  The spatial size is static, as in the reference (generator_with_attention.py:15 reads it from get_shape())."""
        assert images.dim() == 4 and images.shape[3] == 3 and images.shape[1] == images.shape[2], "images must be NHWC [B,S,S,3]"
        B, S = int(images.shape[0]), int(images.shape[1])
        if self.net is None:
            K = kernels_for(images.device)
            E = EMBED_DIM if self.embedding_matrix is None else int(self.embedding_matrix.shape[1])
            sd = init_state_dict(self.kind, self.vocab_size, S, E, seed=self.init_seed)
            if self.kind == "D" and self.embedding_matrix is not None:
                sd["W"] = torch.as_tensor(self.embedding_matrix).detach().float().cpu()
            self.net = Network(K, self.kind, self.vocab_size, S, B, E, state_dict=sd)
            self._nets[B] = self.net
            if self.kind == "D":
                self.embedding_matrix = self.net.arena.views["W"]
        if self.net.trunk.S != S:
            raise ValueError("network was built for %dx%d images; got %dx%d (the attention weights depend on the feature-map size)"
                             % (self.net.trunk.S, self.net.trunk.S, S, S))
        net = self._nets.get(B)
        if net is None:
            for b in [b for b, n in self._nets.items() if n is not self.net and n is not self._last]:
                if len(self._nets) < self.max_cached_batch_sizes:
                    break
                del self._nets[b]         # least recently used first
            net = self._nets[B] = Network(self.net.K, self.kind, self.vocab_size, S, B, self.net.arena.E, share=self.net)
        self._nets.move_to_end(B)
        net.finish_update()          # (a deferred optimiser step of a data-parallel run is applied before the weights are read)
        self._last = net
        return net

    def _publish(self, ctx, st):
        B, L = ctx.shape[0], ctx.shape[1]
        side = int(round(L ** 0.5))
        self._ctx = ctx
        self.downsampled = ctx.view(B, side, side, FEAT_C)
        self.flattened_context = ctx.view(B, L * FEAT_C)
        self.partially_flattened_context = ctx
        self.alpha = st.AL[-1][0]

    def _attention(self, cell_state):
        """attentionMechanism: z_hat for an arbitrary (c, h) state on the current feature map."""
        net, ctx = self._last, self._ctx
        B, L = ctx.shape[0], ctx.shape[1]
        c = cell_state[0].contiguous()
        K = net.K
        ec = torch.empty((1, B, L), device=c.device)
        al = torch.empty((1, B, L), device=c.device)
        z = torch.empty((1, B, FEAT_C), device=c.device)
        K.gemm_nn(c, net.head.W_c, ec[0])
        K.attn_step_fwd(net.head.P, ec, ctx, al, z)
        self.alpha = al[0]
        return z[0]

    def state_dict(self, full_names=True):
        return self.net.arena.state_dict(full_names)

    def load_state_dict(self, sd):
        self.net.arena.load_state_dict(sd)
        self.net.trunk.refresh_weights()
