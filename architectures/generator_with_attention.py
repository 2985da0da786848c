"""Drop-in for the reference's architectures/generator_with_attention.py (class Generator), MI355X-native.

Same import path, constructor and method signatures as the reference (generator_with_attention.py:8-91):
    Generator(vocab_size)
    Generator.build_generator(images, is_training=True) -> logits [B, 3, vocab]      (raw logits, no softmax)
    Generator.attentionMechanism(cell_state)            -> z_hat  [B, 512]           (cell_state = (c, h), c is used)
    attributes after a build: downsampled, flattened_context, partially_flattened_context, alpha
In the reference these methods add TensorFlow ops to a graph; here they run eagerly on the GPU: every arithmetic
op is a hand-written HIP kernel behind the C ABI of libsgg_hip.so (include/sgg_hip.h).  Repeated builds share one
set of weights, as `reuse=tf.AUTO_REUSE` does (train.py:86).  `is_training` is accepted and ignored, as in the
reference (no dropout / batch-norm).  There is no CPU fallback.
"""
import os
import sys

sys.path.append(os.getcwd())
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import torch  # noqa: E402

import sgg_amd  # noqa: E402,F401
from sgg_amd.api import NetworkHandle  # noqa: E402


class Generator(NetworkHandle):

    def __init__(self, vocab_size):
        NetworkHandle.__init__(self, "G", vocab_size)

    def attentionMechanism(self, cell_state):
        return self._attention(cell_state)

    def build_generator(self, images, is_training=True, noise=None):
        """images: float32 NHWC [B, S, S, 3], already standardised (train.py:172).  `noise` [B, 512] replaces the
        in-graph tf.random_normal of the reference (generator_with_attention.py:81); drawn with torch.randn if None."""
        net = self._ensure(images)
        if noise is None:
            noise = torch.randn((images.shape[0], 512), device=images.device, dtype=torch.float32)
        ctx = net.trunk.forward(images.contiguous())
        net.head.precompute(ctx)
        st = net.head.state(1, images.shape[0])
        net.head.forward(st, ctx, noise)
        self._publish(ctx, st)
        return st.OUT[0]
