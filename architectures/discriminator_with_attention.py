"""Drop-in for the reference's architectures/discriminator_with_attention.py (class Discriminator), MI355X-native.

Same import path, constructor and method signatures as the reference (discriminator_with_attention.py:7-93):
    Discriminator(vocab_size, embedding_matrix)
    Discriminator.build_discriminator(input_triples, images, is_training=True) -> critic logits [B, 3, 1]
    Discriminator.attentionMechanism(cell_state) -> z_hat [B, 512]
`input_triples` is float32 [B, 3, vocab]: one-hot real triples or raw generator logits (train.py:173, 242).
`embedding_matrix` [vocab, 300] is created by the trainer and trained by the critic's optimiser
(train.py:68-72, 263); here its storage moves into the critic's parameter arena and `self.embedding_matrix`
is the live view of it.  Every arithmetic op is a HIP kernel behind libsgg_hip.so; no CPU fallback.
"""
import os
import sys

sys.path.append(os.getcwd())
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import sgg_amd  # noqa: E402,F401
from sgg_amd.api import NetworkHandle  # noqa: E402


class Discriminator(NetworkHandle):

    def __init__(self, vocab_size, embedding_matrix):
        NetworkHandle.__init__(self, "D", vocab_size)
        self.embedding_matrix = embedding_matrix

    def attentionMechanism(self, cell_state):
        return self._attention(cell_state)

    def build_discriminator(self, input_triples, images, is_training=True):
        net = self._ensure(images)
        B = images.shape[0]
        ctx = net.trunk.forward(images.contiguous())
        net.head.precompute(ctx)
        st = net.head.state(1, B, "api")
        net.head.forward(st, ctx, [input_triples.contiguous()])
        self._publish(ctx, st)
        return st.OUT[0]
