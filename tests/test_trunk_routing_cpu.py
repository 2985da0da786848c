"""Which weight-fragment layout the encoder asks for (trunk._query_layouts): the producer / consumer 3x3 kernel (layout 4) for every dgrad
and every forward WITHOUT an LN prologue, the four-wave kernel's fragments (layout 1) for the forward of a layer that will run with the
prologue - host logic only, a stub stands in for the kernel library."""
import sgg_amd  # noqa: F401
from sgg_amd import trunk as T


class _K:
    conv_precision = 2
    ln_fusion = 1
    ln_fusion_skip = ()

    def conv_wsplit_layout(self, k, s, H, W, cin, cout):      # what csrc/conv_gather.hip: sgg_conv_wsplit_layout answers in mode 2
        if k == 3 and s == 1 and cout % 128 == 0 and cin % 64 == 0:
            return 4
        return 1 if k == 3 and s == 1 else 2

    def ln_prologue_ok(self, *a):
        return True


def _lay(i, cin, cout, k, s, hw, prev, has_ln=True, region=None):
    return {"i": i, "cin": cin, "cout": cout, "k": k, "s": s, "hin": hw, "win": hw, "has_ln": has_ln, "region": region, "prev": prev,
            "out_shape": (64, hw // s, hw // s, cout)}


def _trunk(K):
    t = T.Trunk.__new__(T.Trunk)
    t.K = K
    return t


def test_forward_with_ln_prologue_keeps_the_four_wave_fragments_and_dgrad_takes_the_pc_kernel():
    K = _K()
    prev = _lay(5, 64, 128, 3, 1, 112, None)                       # conv2_3: LayerNorm output feeds conv2_4
    lay = _lay(6, 128, 128, 3, 1, 112, prev)                      # conv2_4
    assert any(T.ln_fusion_pays(prev["out_shape"], lay["cout"]))   # (the cost model fuses this LayerNorm at batch 64)
    t = _trunk(K)
    t._query_layouts(lay)
    assert lay["ws_layout"] == 1 and lay["ws_layout_bwd"] == 4


def test_forward_without_prologue_takes_the_pc_kernel():
    for kw in ({"has_ln": False}, {"region": (1, 1, 110, 110)}):
        K = _K()
        prev = _lay(5, 64, 128, 3, 1, 112, None, **kw)
        lay = _lay(6, 128, 128, 3, 1, 112, prev)
        _trunk(K)._query_layouts(lay)
        assert lay["ws_layout"] == 4 and lay["ws_layout_bwd"] == 4, kw
    K = _K()
    K.ln_fusion = 0                                               # prologue schedule off
    lay = _lay(6, 128, 128, 3, 1, 112, _lay(5, 64, 128, 3, 1, 112, None))
    _trunk(K)._query_layouts(lay)
    assert lay["ws_layout"] == 4
    K = _K()
    K.ln_fusion_skip = (5,)                                       # this LayerNorm excluded from the schedule
    lay = _lay(6, 128, 128, 3, 1, 112, _lay(5, 64, 128, 3, 1, 112, None))
    _trunk(K)._query_layouts(lay)
    assert lay["ws_layout"] == 4


def test_presplit_activations_keep_the_pc_kernel_and_drop_the_prologue_for_its_consumers():
    """With pre-split activations (K.presplit) a 128-column consumer stages its patch by LDS-DMA on the producer / consumer kernel:
    its producer's LayerNorm keeps the apply pass (trunk._pc_presplit), so the forward asks for layout 4 as well."""
    K = _K()
    K.presplit = True
    prev = _lay(5, 64, 128, 3, 1, 112, None)
    lay = _lay(6, 128, 128, 3, 1, 112, prev)
    t = _trunk(K)
    assert t._pc_presplit(lay) and not t._ln_prologue_expected(lay)
    t._query_layouts(lay)
    assert lay["ws_layout"] == 4 and lay["ws_layout_bwd"] == 4
    K.conv_precision = 3                                          # bf16 pieces: no pre-split tensors, the round-3 routing stands
    t._query_layouts(lay)
    assert lay["ws_layout"] == 1
    lay64 = _lay(4, 64, 64, 3, 1, 112, _lay(3, 32, 64, 3, 1, 112, None))      # a 64-column consumer is not on that kernel
    K.conv_precision = 2
    assert not t._pc_presplit(lay64)


def test_other_layers_are_untouched():
    K = _K()
    lay = _lay(3, 32, 64, 3, 1, 112, _lay(2, 32, 32, 5, 2, 224, None))      # conv2_1: 64 columns
    _trunk(K)._query_layouts(lay)
    assert lay["ws_layout"] == 1 and lay["ws_layout_bwd"] == 1
