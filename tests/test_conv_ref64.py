"""CPU: the per-tap fp64 references of tests/conv_ref64.py (used by the full-size GPU parity tests) agree with the oracle's
convolution and its autograd gradients (oracle/kernels_ref.py), including the asymmetric SAME pads (1,2) and odd sizes."""
import pytest
import torch

from oracle.kernels_ref import RefKernels
from tests import conv_ref64 as R64

CASES = [(2, 12, 12, 8, 16, 3, 1), (2, 12, 10, 8, 8, 5, 2), (1, 13, 11, 4, 8, 5, 2), (2, 7, 9, 3, 8, 3, 1), (1, 8, 8, 16, 8, 5, 2)]


@pytest.mark.parametrize("case", CASES)
def test_tap_matmul_reference_equals_oracle(case):
    B, H, W, Ci, Co, k, s = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, H, W, Ci), generator=g, dtype=torch.float64)
    w = torch.randn((k, k, Ci, Co), generator=g, dtype=torch.float64)
    b = torch.randn((Co,), generator=g, dtype=torch.float64)
    Ho, Wo = R64.same_pads(H, k, s)[0], R64.same_pads(W, k, s)[0]
    dy = torch.randn((B, Ho, Wo, Co), generator=g, dtype=torch.float64)
    ref = RefKernels()
    y = torch.empty((B, Ho, Wo, Co), dtype=torch.float64)
    ref.conv_fwd(x, w, None, b, y, s)
    dw = torch.empty_like(w)
    ref.conv_wgrad(x, dy, dw, s)
    dx = torch.empty_like(x)
    ref.conv_dgrad(dy, w, dx, s)
    assert float((R64.conv_fwd64(x, w, b, s) - y).abs().max()) < 1e-12
    assert float((R64.conv_wgrad64(x, dy, k, s) - dw).abs().max()) < 1e-11
    assert float((R64.conv_dgrad64(dy, w, (H, W), s) - dx).abs().max()) < 1e-12
