"""scene-graph-gan_amd: MI355X-native (gfx950) implementation of the Scene-Graph-GAN WGAN-GP training hot path.

Imported as `sgg_amd` (see ../sgg_amd.py).  Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/sgg_hip.h)
  build.py   hipcc build of libsgg_hip.so
  lib.py     ctypes binding (tensor-level wrapper of the C ABI; no CPU fallback)
  params.py  parameter arenas with the reference's TF variable names
  trunk.py   conv encoder forward / backward orchestration
  head.py    attention + LN-LSTM head forward / backward (fp32 and dual-number passes)
  step.py    WGAN-GP critic step / generator step, TF-Adam
  dp.py      data-parallel gradient all-reduce over RCCL with compute overlap
"""
__version__ = "0.1.0"
