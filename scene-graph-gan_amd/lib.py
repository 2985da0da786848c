"""ctypes binding of libsgg_hip.so (include/sgg_hip.h) on torch device tensors.

PyTorch is plumbing only here: it owns device memory and the stream; every arithmetic op of the hot path is a
hand-written HIP kernel behind the C ABI.  There is NO fallback: if the shared object is missing, fails to
load, or a tensor is not on a HIP device, the call raises.

Tensor conventions used by the host orchestration (trunk.py / head.py / step.py):
  * head tensors carry a leading "plane" dimension: [1, R, W] for plain fp32, [2, R, W] for dual numbers
    (plane 0 = real part, plane 1 = dual part; see csrc/dual.h).  Column slices of such buffers are passed as
    strided views; the binding extracts base pointers and the row stride (leading dimension).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsgg_hip.so")

_vp, _i, _f, _sz, _ll = c_void_p, c_int, c_float, c_size_t, c_longlong

# name -> (restype, argtypes); must list every symbol declared in include/sgg_hip.h
SIGNATURES = {
    "sgg_version": (_i, []),
    "sgg_last_error": (c_char_p, []),
    "sgg_device_info": (_i, [_vp, _vp, _vp, _vp, _i]),
    "sgg_hwio_to_hwoi": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sgg_conv_split_weights": (_i, [_vp, _vp, _ll, _i, _vp, _vp]),
    "sgg_absmax": (_i, [_vp, _ll, _vp, _vp]),
    "sgg_conv_wsplit_layout": (_i, [_i] * 8),
    "sgg_conv_wsplit_layout_presplit": (_i, [_i] * 8),
    "sgg_conv_s2d_weights": (_i, [_vp, _vp, _i, _i, _vp]),
    "sgg_conv_prepare_weights": (_i, [_vp, _i, _i, _vp]),
    "sgg_conv_split_weights_frag": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sgg_conv_split_weights_frag16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sgg_conv2d_nhwc_fwd": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 14 + [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgg_conv2d_nhwc_fwd_tile_stats": (_i, [_i] * 9),
    "sgg_presplit16": (_i, [_vp, _vp, _ll, _vp, _vp]),
    "sgg_conv2d_nhwc_dgrad": (_i, [_vp, _vp, _vp, _vp] + [_i] * 14 + [_vp, _vp, _i, _vp]),
    "sgg_conv2d_nhwc_wgrad_workspace_bytes": (_sz, [_i] * 9),
    "sgg_conv2d_nhwc_wgrad": (_i, [_vp, _vp, _vp] + [_i] * 14 + [_vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sgg_conv2d_nhwc_wgrad_resident": (_i, [_i] * 9),
    "sgg_layernorm_hwc_elu_workspace_bytes": (_sz, [_i, _i, _i]),
    "sgg_layernorm_hwc_elu_fwd": (_i, [_vp] * 7 + [_i] * 10 + [_vp, _sz, _vp]),
    "sgg_layernorm_hwc_elu_bwd": (_i, [_vp] * 10 + [_i] * 9 + [_vp, _vp, _sz, _vp]),
    "sgg_layernorm_hwc_bwd_finalize": (_i, [_vp, _i, _vp]),
    "sgg_layernorm_hwc_finalize": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sgg_spatial_mean_fwd": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "sgg_spatial_mean_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "sgg_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "sgg_gemm_skinny_fwd": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "sgg_gemm_skinny_dgrad": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "sgg_gemm_skinny_wgrad": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "sgg_layernorm_hwc_elu_bwd_sums": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_conv2d_nhwc_wgrad_c3_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_attn_ctx_gemm_fwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgg_attn_ctx_gemm_dgrad": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sgg_attn_ctx_gemm_wgrad": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sgg_attn_step_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sgg_attn_step_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sgg_lnlstm_gates_fwd": (_i, [_vp] * 9 + [_i, _i, _vp]),
    "sgg_lnlstm_gates_bwd": (_i, [_vp] * 7 + [_i] + [_vp] * 7 + [_i, _vp]),
    "sgg_colsum_workspace_bytes": (_sz, [_i, _i]),
    "sgg_colsum": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "sgg_onehot": (_i, [_vp, _vp, _i, _i, _vp]),
    "sgg_resize_bilinear_tf1": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sgg_embed_gather_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "sgg_embed_gather_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sgg_interpolate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sgg_wgan_gp_loss_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sgg_wgan_gp_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sgg_wgan_losses": (_i, [_vp, _vp, _f, _i, _i, _i, _vp, _vp]),
    "sgg_adam_tf_multi": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _vp]),
    "sgg_argmax_rows": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sgg_fill": (_i, [_vp, _ll, _f, _vp]),
}

_lib = None


class ConvWeightDesc(ctypes.Structure):
    """sgg_conv_weight_desc of include/sgg_hip.h."""
    _fields_ = [("w", c_void_p), ("w_hwoi", c_void_p), ("w3", c_void_p), ("w3_hwoi", c_void_p), ("ws_fwd", c_void_p),
                ("ws_bwd", c_void_p), ("amax", c_void_p), ("taps", c_int), ("cin", c_int), ("cout", c_int), ("layout_fwd", c_int),
                ("layout_bwd", c_int)]


class LnFinalizeDesc(ctypes.Structure):
    """sgg_ln_finalize_desc of include/sgg_hip.h."""
    _fields_ = [("workspace", c_void_p), ("gamma", c_void_p), ("stats", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("dbias_prev", c_void_p), ("B", c_int), ("HW", c_int), ("C", c_int), ("HW_valid", c_int)]


class SggError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load libsgg_hip.so and bind every declared symbol. Raises if the library or a symbol is missing.
    SGG_HIP_LIB overrides the path (instrumented builds, scripts/build_prof_lib.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SGG_HIP_LIB", path)
    if not os.path.exists(path):
        raise SggError("libsgg_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def same_pads(in_size: int, k: int, s: int):
    """TF SAME padding (tf.layers.conv2d(padding="same"), generator_with_attention.py:29): (out, before, after)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def _p(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    """row stride of a 2-D view whose last dim is contiguous (torch reports arbitrary strides for size-1 dims)"""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# Documented run-time options of HipKernels (the whole experiment surface of the host side; everything else is a build-time
# -D switch of the library).  HipKernels(options={...}) takes any subset; scripts may override them without
# touching code through ONE environment variable, SGG_OPTIONS="key=value,key=value" (lists: values joined by '+').
DEFAULT_OPTIONS = {
    # Convolution contraction mode (csrc/conv_gather.hip, conv_wgrad.hip):
    #   2 (default) f32 operands scaled by a per-tensor power of two and split into two fp16 pieces with round-to-nearest
    #               (23 significant bits), 3 fp16 MFMAs per product, f32 accumulate: error against fp64 within 2x the native
    #               f32 path's + 3e-7 (tests/test_fullsize_conv_gpu.py);
    #   6           three bf16 pieces, 6 bf16 MFMAs (no scaling needed): also f32-equivalent, slower;
    #   0           native f32 MFMA (v_mfma_f32_32x32x2_f32, bit-exact f32 fmaf chain);
    #   3           two bf16 pieces, 3 MFMAs: drops 2^-17 cross terms (inside the stated 1e-4 tolerance);
    #   1 / 4       ONE fp16 / bf16 piece: mixed-precision modes, NOT the reference's arithmetic (include/sgg_hip.h).
    "conv_precision": 2,
    # False: every convolution on the gather kernels (no resident halo / band / producer-consumer kernels)
    "conv_halo": True,
    # False: the 128-column 3x3 layers stay on the four-wave halo kernel (w_split_layout 1 instead of 4)
    "halo_pc": True,
    # True: the dgrads of the 64-column 3x3 layers whose dy arrives pre-split (conv2_2, conv2_3) run on the four-block form of the
    # producer / consumer kernel (trunk._query_layouts; csrc/conv_halo_pc.hip NB = 4)
    "halo_pc64": True,
    # LayerNorm + ELU applied by the consuming convolution's patch staging (LN prologue): 1 (default) = per layer and per KIND of
    # encoder pass (forward-only / followed by a backward) where the measured cost models (trunk.ln_fusion_pays, pc_ln_fusion_pays) say it pays
    # (trunk._plan_ln_fusion: with pre-split activations 20 of the 44 apply passes of a step at configs[1] run as prologues, 24 as
    # standalone passes - DESIGN.md "The LN prologue"); 2 = wherever the kernels allow (slower: DESIGN.md); 0 = never
    "ln_fusion": 1,
    # True: the LayerNorm kernels write their outputs pre-split for the convolutions that consume them (trunk._plan_s16; fp16 modes)
    "presplit": True,
    # True (with presplit): the gradient the attention head hands to the last convolution is converted to the pre-split format once
    # per backward (sgg_presplit16), so that `downsampled`'s dgrad and filter gradient stage it by DMA as well
    "presplit_head_grad": True,
    # two-stream schedule: True = the filter gradient of layer j starts when dgrad_j has finished (beside the LayerNorm backward of
    # layer j - 1) instead of together with dgrad_j (trunk.backward)
    "wgrad_late": True,
    # two-stream schedule: HIP priority of the side streams (0 = default; positive = lower than the main stream's, negative = higher)
    "side_priority": 0,
    # two-stream schedule: 1 = G's encoder forward of the update after a critic update starts on a stream of its own as soon as that
    # critic update's G head has run, beside its heads and encoder backward (step.GanStep._g_early_stream)
    "g_early": 1,
    # ... and while it runs there its persistent convolution kernels occupy only this many of an XCD's 32 CUs (0 = all): the critic's
    # recurrent heads - a chain of short launches on the critical path - then find free CUs at once instead of waiting for a resident
    # workgroup of G's forward to finish its tile.  The convolutions lose nothing on 28 of 32 CUs (every persistent kernel capped at
    # 28: +0.25 ms per step only, profiles/r05_persistent_cu_cap_ab.log); 28 here: 42.75 / 42.80 against 43.04 / 42.97 ms per step, on
    # another box 44.02 / 44.07 against 44.32 / 44.18; 30, 26, 24, 20: equal to none (profiles/r05_early_forward_cu_cap_ab*.log).
    # Tiles, products and summation orders are unchanged: bit-identical results (tests/test_concurrency_gpu.py)
    "g_early_cus": 28,
    # True: the LayerNorm backward of conv1_1's output runs without its apply pass - conv1_1's filter gradient, the only consumer of
    # that dy, computes it itself (sgg_conv2d_nhwc_wgrad_c3_ln): one read of y and da instead of the apply pass (2 reads + 1 write of
    # 411 MB at batch 64) + the filter gradient's read of dy, on the tail of every encoder backward
    "c3_ln_bwd_fused": True,
    "d_side_cus": 0,      # the same for D's encoder forward on the side stream (beside G's forward and G's head)
    "fwd_cus": 0,         # the same for every encoder forward (16: G's and D's forwards of an update on disjoint halves of the chip)
    # cost-model overrides for A/B runs (conv indices): never fuse / fuse in forward-only passes / fuse in passes with backward /
    # never fuse in passes with backward
    "ln_fusion_skip": (),
    "ln_fusion_force": (),
    "ln_fusion_force_bwd": (),
    "ln_fusion_skip_bwd": (),
}


def options_from_env(base=None):
    """DEFAULT_OPTIONS (or `base`) with the overrides of SGG_OPTIONS applied: the one environment hook of the host side."""
    opts = dict(DEFAULT_OPTIONS if base is None else base)
    for item in filter(None, os.environ.get("SGG_OPTIONS", "").split(",")):
        key, _, val = item.partition("=")
        key = key.strip()
        if key not in DEFAULT_OPTIONS:
            raise SggError("SGG_OPTIONS: unknown option %r (known: %s)" % (key, ", ".join(sorted(DEFAULT_OPTIONS))))
        d = DEFAULT_OPTIONS[key]
        if isinstance(d, bool):
            opts[key] = val.strip().lower() not in ("0", "false", "no", "off", "")
        elif isinstance(d, tuple):
            opts[key] = tuple(int(v) for v in val.split("+") if v.strip())
        else:
            opts[key] = int(val)
    return opts


class HipKernels:
    """Tensor-level wrapper of the C ABI. All tensors must be fp32 (int64 where stated) on one HIP device.
    options: a subset of DEFAULT_OPTIONS; they become attributes of the same name (read by trunk.py)."""

    name = "hip"

    def __init__(self, device=None, options=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise SggError("no HIP device visible: the scene-graph-gan_amd product path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._ws_by_stream = {}
        self.timing = None      # bench.py sets this to a list: launches are then bracketed by HIP events
        self.timing_conv_only = False   # True: only the MFMA-bound convolution calls are bracketed (the timed region of bench.py)
        self.timing_symbols = None      # with timing_conv_only True: bracket only these kernel symbols (bench.py: the dominant one)
        opts = options_from_env()
        for key, val in (options or {}).items():
            if key not in DEFAULT_OPTIONS:
                raise SggError("HipKernels: unknown option %r" % key)
            opts[key] = val
        for key, val in opts.items():
            setattr(self, key, tuple(val) if isinstance(DEFAULT_OPTIONS[key], tuple) else val)
        self.conv_precision = int(self.conv_precision)
        assert self.conv_precision in (0, 1, 2, 3, 4, 6)       # 1 / 4: single-piece (mixed-precision) modes, include/sgg_hip.h
        self._amax_by_stream = {}

    def _timed(self, symbol, flops, fn, nbytes=0.0):
        """Run fn() between two HIP events on the launch stream when kernel timing is on (bench.py roofline legs).
        flops / nbytes: ALGORITHMIC work of the call (MFMA-bound convs: flops; HBM-bound kernels: bytes)."""
        # timing_conv_only: True = the forward / dgrad convolution launches only (the candidates of bench.py's `roofline`: every event
        # pair costs ~5 us of the timed region), "mfma" = every call with algorithmic FLOPs, False = everything
        if self.timing is None or (self.timing_conv_only == "mfma" and flops <= 0.0) or \
                (self.timing_conv_only is True and (not symbol.startswith(("conv_halo", "conv_s2", "conv_gather")) or
                                                   (self.timing_symbols is not None and symbol not in self.timing_symbols))):
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.device))
        r = fn()
        e1.record(torch.cuda.current_stream(self.device))
        self.timing.append((symbol, flops, nbytes, e0, e1))
        return r

    def gather_symbol(self, n_out, presplit=False):
        """Kernel symbol (as rocprofv3 prints it, spaces removed) that csrc/conv_gather.hip: dispatch_gather picks."""
        if self.conv_precision:
            tile = "128,128,2,2" if n_out % 128 == 0 else ("256,64,4,1" if n_out % 64 == 0 else "256,32,4,1")
            return "conv_gather_bf16s_kernel<%s,%d,%s,%s,32>" % (tile, 3 if self.conv_precision == 6 else 2,
                                                                "true" if presplit else "false",
                                                                "true" if self.conv_precision in (1, 2) else "false")
        if n_out % 128 == 0:
            return "conv_gather3_kernel<128,128,2,2,32>"
        return "conv_gather_kernel<256,64,4,1>" if n_out % 64 == 0 else "conv_gather3_kernel<256,32,4,1,32>"

    # -- plumbing ------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check(self, rc, name):
        if rc != 0:
            raise SggError("%s failed (%d): %s" % (name, rc, self.lib.sgg_last_error().decode()))

    def _dev(self, *ts):
        for t in ts:
            if t is not None and (not t.is_cuda):
                raise SggError("tensor is not on a HIP device (no CPU fallback in the product path)")

    def workspace(self, nbytes: int):
        """Scratch buffer of the CURRENT stream (kernels on different streams may run concurrently)."""
        sid = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws_by_stream.get(sid)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(int(nbytes * 1.25) + 256, 1 << 20), dtype=torch.uint8, device=self.device)
            self._ws_by_stream[sid] = ws
        return ws

    def device_info(self):
        cu, lds, hbm = c_int(0), c_size_t(0), c_size_t(0)
        arch = ctypes.create_string_buffer(64)
        self._check(self.lib.sgg_device_info(ctypes.addressof(cu), ctypes.addressof(lds), ctypes.addressof(hbm),
                                             ctypes.addressof(arch), 64), "sgg_device_info")
        return {"cu_count": cu.value, "lds_bytes_per_cu": lds.value, "hbm_bytes": hbm.value, "arch": arch.value.decode()}

    # -- conv encoder ----------------------------------------------------------------------------------
    def hwio_to_hwoi(self, w, wt):
        self._dev(w, wt)
        kh, kw, ci, co = w.shape
        self._check(self.lib.sgg_hwio_to_hwoi(_p(w), _p(wt), kh * kw, ci, co, self._stream()), "sgg_hwio_to_hwoi")

    @staticmethod
    def _conv_dims(x_shape, w_shape, stride):
        B, Hi, Wi, Ci = x_shape
        KH, KW, Ci2, Co = w_shape
        assert Ci == Ci2
        Ho, pt, _ = same_pads(Hi, KH, stride)
        Wo, pl, _ = same_pads(Wi, KW, stride)
        return B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pt, pl

    def absmax(self, x, amax):
        """amax (1-element fp32 view, zeroed by the caller) = max(amax, max|x|)."""
        self._dev(x, amax)
        assert x.is_contiguous()
        self._check(self.lib.sgg_absmax(_p(x), x.numel(), _p(amax), self._stream()), "sgg_absmax")

    def presplit16(self, x, out, amax):
        """out = x in the pre-split format of the LayerNorm kernels (ln_elu_fwd(..., out_s16=True)) under the scale of the word `amax`
        (max|x| or a bound); x: contiguous f32 NHWC with C % 32 == 0; out: same shape (may be x)."""
        self._dev(x, out, amax)
        assert x.is_contiguous() and out.is_contiguous() and out.numel() == x.numel() and x.shape[-1] % 32 == 0
        self._check(self.lib.sgg_presplit16(_p(x), _p(out), x.numel(), _p(amax), self._stream()), "sgg_presplit16")

    def _amax_or_compute(self, t, amax, slot):
        """precision 2 needs max|t| on the device; callers that track it pass `amax`, otherwise it is computed here."""
        if self.conv_precision not in (1, 2) or amax is not None or t is None:
            return amax
        sid = torch.cuda.current_stream(self.device).cuda_stream     # scratch words per stream, like workspace()
        scratch = self._amax_by_stream.get(sid)
        if scratch is None:
            scratch = self._amax_by_stream[sid] = torch.zeros(8, dtype=torch.float32, device=self.device)
        w = scratch[slot:slot + 1]
        self.fill(w, 0.0)
        self.absmax(t, w)
        return w

    def conv_wsplit_layout(self, k, stride, H, W, cin, cout):
        """Layout the conv entry points want for the pre-split weights of this layer: 0 = planes, 1 / 2 / 3 = MFMA fragment order
        (1: halo-resident 3x3 stride-1 kernel, 2: band-resident 5x5 stride-2 kernel, 3: conv1_3 through the space-to-depth view;
        H, W = the full-resolution grid), 4 = the fragments of the K = 32 MFMA shape for the producer / consumer 3x3 kernel
        (128-column tiles); SGG_CONV_HALO=0 keeps every layer on the gather kernel."""
        if not self.conv_halo:
            return 0
        lay = self.lib.sgg_conv_wsplit_layout(k, k, stride, H, W, cin, cout, self.conv_precision)
        return 1 if (lay == 4 and not self.halo_pc) else lay

    def conv_wsplit_layout_presplit(self, k, stride, H, W, cin, cout):
        """conv_wsplit_layout for a launch whose source operand WILL be a pre-split tensor: the 64-column 3x3 layers then also take
        layout 4 (the four-block form of the producer / consumer kernel; option halo_pc64).  Such a launch must pass x_s16 / dy_s16."""
        if not self.conv_halo:
            return 0
        if not (self.halo_pc and getattr(self, "halo_pc64", True)):
            return self.conv_wsplit_layout(k, stride, H, W, cin, cout)
        return self.lib.sgg_conv_wsplit_layout_presplit(k, k, stride, H, W, cin, cout, self.conv_precision)

    def split_weights(self, w, out, amax=None, layout=0):
        """w fp32 [kh, kw, N, C] -> out int16 [P, n] sixteen-bit planes (layout 0) or MFMA B fragments (layout 1)."""
        self._dev(w, out, amax)
        amax = self._amax_or_compute(w, amax, 2)
        if layout == 4:         # fragments of the K = 32 MFMA shape (producer / consumer 3x3 kernel)
            kh, kw, n, c = w.shape
            self._check(self.lib.sgg_conv_split_weights_frag16(_p(w), _p(out), kh * kw, n, c, self.conv_precision, _p(amax),
                                                               self._stream()), "sgg_conv_split_weights_frag16")
            return
        if layout in (1, 2, 3):
            kh, kw, n, c = w.shape
            self._check(self.lib.sgg_conv_split_weights_frag(_p(w), _p(out), kh * kw, n, c, self.conv_precision, _p(amax),
                                                             self._stream()), "sgg_conv_split_weights_frag")
            return
        self._check(self.lib.sgg_conv_split_weights(_p(w), _p(out), w.numel(), self.conv_precision, _p(amax), self._stream()),
                    "sgg_conv_split_weights")

    def s2d_weights(self, w5, w3):
        """HWIO [5,5,Ci,Co] -> the 9-tap kernel [3,3,4*Ci,Co] of the same stride-2 convolution over the space-to-depth view of x
        (w_split_layout 3; include/sgg_hip.h)."""
        self._dev(w5, w3)
        kh, kw, ci, co = w5.shape
        assert (kh, kw) == (5, 5) and tuple(w3.shape) == (3, 3, 4 * ci, co) and w5.is_contiguous() and w3.is_contiguous()
        self._check(self.lib.sgg_conv_s2d_weights(_p(w5), _p(w3), ci, co, self._stream()), "sgg_conv_s2d_weights")

    def weight_descs(self, layers):
        """layers: dicts with w, w_fwd, (w3, w3_fwd), ws_fwd, ws_bwd, amax (1-word view or None), ws_layout, ws_layout_bwd ->
        a ctypes array of sgg_conv_weight_desc for prepare_weights (the tensors must stay alive and in place)."""
        arr = (ConvWeightDesc * len(layers))()
        for d, lay in zip(arr, layers):
            w = lay["w"]
            kh, kw, ci, co = w.shape
            self._dev(w, lay["w_fwd"], lay.get("w3"), lay.get("w3_fwd"), lay.get("ws_fwd"), lay.get("ws_bwd"), lay.get("amax"))
            d.w, d.w_hwoi = _p(w), _p(lay["w_fwd"])
            d.w3, d.w3_hwoi = _p(lay.get("w3")), _p(lay.get("w3_fwd"))
            d.ws_fwd, d.ws_bwd, d.amax = _p(lay.get("ws_fwd")), _p(lay.get("ws_bwd")), _p(lay.get("amax"))
            d.taps, d.cin, d.cout, d.layout_fwd, d.layout_bwd = kh * kw, ci, co, lay["ws_layout"], lay["ws_layout_bwd"]
        return arr

    def prepare_weights(self, descs):
        """Transposes, max|w|, space-to-depth kernels and both pre-split copies of every layer in `descs` (weight_descs): three
        launches for a whole encoder (sgg_conv_prepare_weights)."""
        self._check(self.lib.sgg_conv_prepare_weights(ctypes.addressof(descs), len(descs), self.conv_precision, self._stream()),
                    "sgg_conv_prepare_weights")

    def conv_tile_stats_count(self, y_shape, cin, k=0, stride=0, layout=0):
        """(count, mean, M2) triples per sample the forward conv emits for this output shape in the current mode (0: none)."""
        return self.lib.sgg_conv2d_nhwc_fwd_tile_stats(y_shape[1], y_shape[2], cin, y_shape[3], k, k, stride, self.conv_precision,
                                                       layout)

    def halo_pc_symbol(self, lnp=False, presplit=False, n_out=128):
        """Kernel symbol of the producer / consumer 3x3 kernel (csrc/conv_halo_pc.hip; w_split_layout 4); presplit: the source is a
        pre-split tensor, the patch is staged by LDS-DMA (third template argument); fourth: 8x8 blocks per workgroup tile (2 blocks x
        128 columns, or 4 x 64 for the 64-column launches)."""
        dma = presplit and not lnp and self.conv_precision == 2
        return "conv_halo3_pc_kernel<%s,%s,%s,%d>" % ("true" if self.conv_precision == 2 else "false", "true" if lnp else "false",
                                                      "true" if dma else "false", 2 if n_out % 128 == 0 else 4)

    def halo_symbol(self, n_out, n_in, lnp=False):
        """Kernel symbol (as rocprofv3 prints it, spaces removed) that csrc/conv_halo.hip: sgg_halo_launch picks (default build)."""
        tile = "2,128,2,2" if n_out % 128 == 0 else ("2,64,2,2" if n_out % 64 == 0 else "2,32,2,1")
        # (last argument: blocks per wave, 1 in the default build - csrc/conv_halo.hip SGG_HALO_N128_WB2)
        return "conv_halo3_kernel<%s,%s,%s,%s,%s,%s,1>" % (tile, "true" if self.conv_precision in (1, 2) else "false", "true",
                                                           "true" if n_in == 32 else "false", "true" if lnp else "false",
                                                           "true" if self.conv_precision in (1, 4) else "false")

    def s2_symbol(self, dgrad, m_positions=1 << 30, n_out=128, stats=True, lnp=False, presplit=False):
        """(csrc/conv_s2.hip: 224-position bands; with at most 256 work items the channel chunks are split over two workgroups);
        presplit: the source is a pre-split tensor, the patch is staged by LDS-DMA (sixth template argument) - by eight-wave
        workgroups that own 256 output columns where the layer has that many (seventh)"""
        mt = 7
        dma = presplit and not lnp and self.conv_precision == 2
        return "conv_s2_kernel<%s,%s,%d,%s,%s,%s,%d>" % ("true" if dgrad else "false", "true" if self.conv_precision in (1, 2) else "false", mt,
                                                         "true" if self.conv_precision in (1, 4) else "false", "true" if lnp else "false",
                                                         "true" if dma else "false",
                                                         8 if dma and n_out % 256 == 0 and -(-m_positions // 224) * (n_out // 256) > 128 else 4)

    def conv_fwd(self, x, w_hwio, w_fwd, bias, y, stride, w_split=None, amax_x=None, amax_w=None, tile_stats=None, w_split_layout=0,
                 ln=None, x_s16=False, cu_cap=0):
        """y = conv2d_same(x, w) + bias. w_fwd: HWOI transpose of w_hwio (or w_hwio itself when Cin == 3).
        ln = (stats [B,2], gamma, beta): x is the producing layer's PRE-LayerNorm output; the kernel applies LN + ELU while staging
        (halo-resident kernel only; amax_x = the word ln_finalize published).
        x_s16: x is a pre-split tensor (ln_elu_fwd(..., out_s16=True); amax_x = the word that call published).
        cu_cap (1 .. 31): the persistent kernels occupy at most that many of an XCD's 32 CUs (operand_format bits 8 .. 13)."""
        self._dev(x, w_fwd, bias, y)
        ln_s, ln_g, ln_b = ln if ln is not None else (None, None, None)
        self._dev(ln_s, ln_g, ln_b)
        d = self._conv_dims(x.shape, w_hwio.shape, stride)
        assert tuple(y.shape) == (d[0], d[4], d[5], d[6]) and x.is_contiguous() and y.is_contiguous()
        flops = 2.0 * d[0] * d[4] * d[5] * d[6] * d[7] * d[8] * d[3]
        sym = "conv_c3_fwd_kernel" if d[3] == 3 else (self.halo_pc_symbol(ln is not None, x_s16, d[6]) if w_split_layout == 4 else
                                                             self.halo_symbol(d[6], d[3], ln is not None) if w_split_layout == 1 else
                                                             self.halo_symbol(d[6], 4 * d[3], ln is not None) if w_split_layout == 3 else
                                                             self.s2_symbol(False, d[0] * d[4] * d[5], d[6], tile_stats is not None, ln is not None, x_s16) if w_split_layout == 2 else
                                                             self.gather_symbol(d[6], w_split is not None))
        nb = 0.0
        if d[3] != 3:
            amax_x, amax_w = self._amax_or_compute(x, amax_x, 0), self._amax_or_compute(w_fwd, amax_w, 1)
        else:
            nb = 4.0 * (x.numel() + y.numel())      # conv1_1 (K = 27) is HBM-bound: the image read once, y written once
        self._check(self._timed(sym, flops, lambda: self.lib.sgg_conv2d_nhwc_fwd(
            _p(x), _p(w_fwd), _p(w_split), _p(bias), _p(y), *d, self.conv_precision, w_split_layout, _p(amax_x), _p(amax_w),
            _p(tile_stats), _p(ln_s), _p(ln_g), _p(ln_b), int(bool(x_s16)) | ((int(cu_cap) & 63) << 8), self._stream()), nb),
            "sgg_conv2d_nhwc_fwd")

    def conv_dgrad(self, dy, w_hwio, dx, stride, w_split=None, amax_dy=None, amax_w=None, w_split_layout=0, dy_s16=False):
        """dy_s16: dy is a pre-split tensor (ln_elu_bwd(..., out_s16=True); amax_dy = the word that call published)."""
        self._dev(dy, w_hwio, dx)
        d = self._conv_dims(dx.shape, w_hwio.shape, stride)
        assert tuple(dy.shape) == (d[0], d[4], d[5], d[6]) and dy.is_contiguous() and dx.is_contiguous()
        flops = 2.0 * d[0] * d[4] * d[5] * d[6] * d[7] * d[8] * d[3]
        amax_dy, amax_w = self._amax_or_compute(dy, amax_dy, 0), self._amax_or_compute(w_hwio, amax_w, 1)
        sym = self.halo_pc_symbol(False, dy_s16, d[3]) if w_split_layout == 4 else self.halo_symbol(d[3], d[6]) if w_split_layout == 1 else (self.halo_symbol(4 * d[3], d[6]) if w_split_layout == 3 else
                                                                          self.s2_symbol(True, d[0] * d[4] * d[5], d[3], False, False, dy_s16) if w_split_layout == 2 else
                                                                          self.gather_symbol(d[3], w_split is not None))
        self._check(self._timed(sym, flops, lambda: self.lib.sgg_conv2d_nhwc_dgrad(
            _p(dy), _p(w_hwio), _p(w_split), _p(dx), *d, self.conv_precision, w_split_layout, _p(amax_dy), _p(amax_w),
            int(bool(dy_s16)), self._stream())), "sgg_conv2d_nhwc_dgrad")

    def wgrad_resident(self, B, Ho, Wo, cin, cout, k, stride):
        """True if conv_wgrad of this shape runs on the halo-resident kernel (the one that takes pre-split operands)."""
        return bool(self.conv_halo and self.lib.sgg_conv2d_nhwc_wgrad_resident(B, Ho, Wo, cin, cout, k, k, stride, self.conv_precision))

    def conv_wgrad(self, x, dy, dw, stride, amax_x=None, amax_dy=None, ln=None, x_s16=False, dy_s16=False):
        """ln: as conv_fwd (x = pre-LayerNorm output of the producing layer; halo-resident wgrad kernel only).
        x_s16 / dy_s16: the operand is a pre-split tensor of ln_elu_fwd / ln_elu_bwd (out_s16=True)."""
        self._dev(x, dy, dw)
        ln_s, ln_g, ln_b = ln if ln is not None else (None, None, None)
        self._dev(ln_s, ln_g, ln_b)
        d = self._conv_dims(x.shape, dw.shape, stride)
        assert tuple(dy.shape) == (d[0], d[4], d[5], d[6]) and x.is_contiguous() and dy.is_contiguous() and dw.is_contiguous()
        need = self.lib.sgg_conv2d_nhwc_wgrad_workspace_bytes(*d[:9])
        ws = self.workspace(need)
        flops = 2.0 * d[0] * d[4] * d[5] * d[6] * d[7] * d[8] * d[3]
        sym, nb = "conv_wgrad(call: wgrad kernel + slab reduce)", 0.0
        if x_s16 and dy_s16 and ln is None and self.conv_precision == 2 and self.conv_halo and \
                self.lib.sgg_conv2d_nhwc_wgrad_resident(d[0], d[4], d[5], d[3], d[6], d[7], d[8], stride, 2) == 2:
            sym = "conv_wgrad_dma(call: LDS-DMA kernel on pre-split operands + slab reduce)"
        if d[3] != 3:
            amax_x, amax_dy = self._amax_or_compute(x, amax_x, 0), self._amax_or_compute(dy, amax_dy, 1)
        else:
            sym, nb = "conv_c3_wgrad(call: kernel + slab reduce)", 4.0 * (x.numel() + dy.numel())      # conv1_1: HBM-bound
        self._check(self._timed(sym, flops, lambda: self.lib.sgg_conv2d_nhwc_wgrad(
            _p(x), _p(dy), _p(dw), *d, self.conv_precision, 0 if self.conv_halo else 1, _p(amax_x), _p(amax_dy), _p(ln_s), _p(ln_g), _p(ln_b),
            int(bool(x_s16)) | (int(bool(dy_s16)) << 1), _p(ws), ws.numel(), self._stream()), nb),
            "sgg_conv2d_nhwc_wgrad")

    @staticmethod
    def _region(region, H, W):
        """(y0, x0, Hv, Wv) valid region of the H x W plane -> the C ABI's (W, y0, x0, Hv, Wv); W = 0: everything is valid."""
        if region is None:
            return (0, 0, 0, 0, 0)
        y0, x0, hv, wv = region
        assert 0 <= y0 and 0 <= x0 and y0 + hv <= H and x0 + wv <= W, (region, H, W)
        return (W, y0, x0, hv, wv)

    def ln_elu_fwd(self, y, gamma, beta, a, stats, amax_out=None, tile_stats=None, region=None, out_s16=False):
        """tile_stats [B, n, 4]: per-tile (count, mean, M2, max dev) written by conv_fwd's epilogue (skips the statistics pass).
        region (y0, x0, Hv, Wv): LayerNorm over that window of every sample only; `a` is written as zeros outside it.
        out_s16: `a` is written pre-split for the consuming convolutions (include/sgg_hip.h, out_format 1); amax_out (zeroed by the
        caller) then receives an upper bound of max|a|."""
        self._dev(y, gamma, beta, a, stats, amax_out, tile_stats)
        B, H, W, C = y.shape
        assert region is None or tile_stats is None
        need = self.lib.sgg_layernorm_hwc_elu_workspace_bytes(B, H * W, C)
        ws = self.workspace(need)
        nts = 0 if tile_stats is None else tile_stats.shape[1]
        # algorithmic bytes: read y (twice without epilogue statistics: statistics pass + apply pass), write a
        nb = 4.0 * y.numel() * (2 if nts else 3)
        self._check(self._timed("ln_elu_fwd(call)", 0.0, lambda: self.lib.sgg_layernorm_hwc_elu_fwd(
            _p(y), _p(gamma), _p(beta), _p(a), _p(stats), _p(amax_out), _p(tile_stats), nts, B, H * W, C,
            *self._region(region, H, W), int(bool(out_s16)), _p(ws), ws.numel(), self._stream()), nb), "sgg_layernorm_hwc_elu_fwd")

    def ln_finalize(self, tile_stats, gamma, beta, stats, amax_out, hw):
        """Statistics only: stats [B,2] = (mean, rstd) from the conv epilogue's tile partials [B,n,4]; amax_out (1 word, may be None)
        is max-ed with an upper bound of max|ELU(LN(y))|.  For consumers with an LN prologue (conv_fwd / conv_wgrad ln=...)."""
        self._dev(tile_stats, gamma, beta, stats, amax_out)
        B, nts, _ = tile_stats.shape
        self._check(self.lib.sgg_layernorm_hwc_finalize(_p(tile_stats), nts, _p(gamma), _p(beta), _p(stats), _p(amax_out), B, hw,
                                                        gamma.shape[0], self._stream()), "sgg_layernorm_hwc_finalize")

    def ln_prologue_ok(self, k, stride, H, W, cin, cout):
        """True if conv_fwd AND conv_wgrad of this layer can apply the producing layer's LayerNorm + ELU themselves
        (halo-resident kernels of the split modes: 3x3 stride 1, and conv1_3 - forward through the space-to-depth view, wgrad in
        its four parity-class launches; the C ABI rejects the prologue elsewhere)."""
        if not (self.conv_halo and self.conv_precision in (2, 3) and cin <= 512):
            return False
        lay = self.lib.sgg_conv_wsplit_layout(k, k, stride, H, W, cin, cout, self.conv_precision)
        # (layout 2: the band-resident forward has the prologue; its wgrad only on grids the 8x8-block halo kernel tiles)
        return lay in (1, 3, 4) or (lay == 2 and (H // 2) % 8 == 0 and (W // 2) % 8 == 0)

    def ln_prologue_fwd_ok(self, k, stride, H, W, cin, cout):
        """True if conv_fwd of this layer can apply the producing layer's LayerNorm + ELU itself (forward-only passes): the
        resident kernels (halo incl. conv1_3 through the space-to-depth view, band-resident 5x5 stride 2)."""
        return (self.conv_halo and self.conv_precision in (2, 3) and cin <= 512 and
                self.lib.sgg_conv_wsplit_layout(k, k, stride, H, W, cin, cout, self.conv_precision) in (1, 2, 3, 4))

    def ln_elu_bwd_sums(self, y, da, gamma, beta, stats, means, ws):
        """The reduction half of ln_elu_bwd (partial sums into the layer's own workspace `ws` for ln_bwd_finalize) + the two per-sample
        means [B,2] of the backward, WITHOUT the apply pass: the consumer computes dy itself (conv_c3_wgrad_ln)."""
        self._dev(y, da, gamma, beta, stats, means, ws)
        B, H, W, C = y.shape
        nb = 4.0 * 2 * y.numel()
        self._check(self._timed("ln_elu_bwd_sums(call)", 0.0, lambda: self.lib.sgg_layernorm_hwc_elu_bwd_sums(
            _p(y), _p(da), _p(gamma), _p(beta), _p(stats), _p(means), None, None, None, B, H * W, C, _p(ws), ws.numel(), self._stream()), nb),
            "sgg_layernorm_hwc_elu_bwd_sums")

    def conv_c3_wgrad_ln(self, x, y, da, gamma, beta, stats, means, dw):
        """conv1_1's filter gradient with dy = LayerNormBackward(y, da) computed inside the kernel (include/sgg_hip.h)."""
        self._dev(x, y, da, gamma, beta, stats, means, dw)
        B, H, W, _ = x.shape
        assert tuple(y.shape) == (B, H, W, 32) and tuple(da.shape) == (B, H, W, 32) and tuple(dw.shape) == (3, 3, 3, 32)
        assert x.is_contiguous() and y.is_contiguous() and da.is_contiguous() and dw.is_contiguous()
        need = self.lib.sgg_conv2d_nhwc_wgrad_workspace_bytes(B, H, W, 3, H, W, 32, 3, 3)
        ws = self.workspace(need)
        flops = 2.0 * B * H * W * 32 * 27
        nb = 4.0 * (x.numel() + y.numel() + da.numel())
        self._check(self._timed("conv_c3_wgrad_ln(call: LN-backward apply fused + slab reduce)", flops, lambda: self.lib.sgg_conv2d_nhwc_wgrad_c3_ln(
            _p(x), _p(y), _p(da), _p(gamma), _p(beta), _p(stats), _p(means), _p(dw), B, H, W, 1, 1, _p(ws), ws.numel(), self._stream()), nb),
            "sgg_conv2d_nhwc_wgrad_c3_ln")

    def ln_workspace_bytes(self, shape):
        B, H, W, C = shape
        return self.lib.sgg_layernorm_hwc_elu_workspace_bytes(B, H * W, C)

    def ln_finalize_descs(self, layers):
        """layers: dicts with ws (the layer's own workspace), gamma, stats, dgamma, dbeta, dbias, shape (B, H, W, C), region ->
        ctypes array for ln_bwd_finalize (the tensors must stay alive and in place)."""
        arr = (LnFinalizeDesc * len(layers))()
        for d, lay in zip(arr, layers):
            B, H, W, C = lay["shape"]
            self._dev(lay["ws"], lay["gamma"], lay["stats"], lay["dgamma"], lay["dbeta"], lay.get("dbias"))
            d.workspace, d.gamma, d.stats = _p(lay["ws"]), _p(lay["gamma"]), _p(lay["stats"])
            d.dgamma, d.dbeta, d.dbias_prev = _p(lay["dgamma"]), _p(lay["dbeta"]), _p(lay.get("dbias"))
            d.B, d.HW, d.C = B, H * W, C
            d.HW_valid = H * W if lay.get("region") is None else lay["region"][2] * lay["region"][3]
        return arr

    def ln_bwd_finalize(self, descs):
        """dgamma, dbeta and the producing convolutions' bias gradients of several LayerNorms whose ln_elu_bwd ran with
        deferred reductions (dgamma=None, ws=...): one launch (sgg_layernorm_hwc_bwd_finalize)."""
        self._check(self.lib.sgg_layernorm_hwc_bwd_finalize(ctypes.addressof(descs), len(descs), self._stream()),
                    "sgg_layernorm_hwc_bwd_finalize")

    def ln_elu_bwd(self, y, da, gamma, beta, stats, dy, dgamma, dbeta, dbias_prev, amax_out=None, region=None, ws=None, out_s16=False,
                   pq=None):
        """dgamma = dbeta = None with a workspace `ws` of the layer's own: the parameter-gradient reductions are deferred to
        ln_bwd_finalize.  out_s16: dy is written pre-split (amax_out receives an upper bound of max|dy|); pq: two ZEROED words for
        the maxima that bound is made of (default: a pair zeroed here - one more launch)."""
        self._dev(y, da, gamma, beta, stats, dy, dgamma, dbeta, dbias_prev, amax_out, ws)
        B, H, W, C = y.shape
        need = self.lib.sgg_layernorm_hwc_elu_workspace_bytes(B, H * W, C)
        assert (dgamma is None) == (dbeta is None) and (dgamma is not None or ws is not None)
        if ws is None:
            ws = self.workspace(need)
        if out_s16 and pq is None:
            pq = torch.zeros(2, dtype=torch.float32, device=y.device)
        self._dev(pq)
        assert ws.numel() * ws.element_size() >= need
        # bytes: the reduction reads y and da, the apply pass reads them again and writes dy
        self._check(self._timed("ln_elu_bwd(call)", 0.0, lambda: self.lib.sgg_layernorm_hwc_elu_bwd(
            _p(y), _p(da), _p(gamma), _p(beta), _p(stats), _p(dy), _p(dgamma), _p(dbeta), _p(dbias_prev), _p(amax_out), B, H * W, C,
            *self._region(region, H, W), int(bool(out_s16)), _p(pq), _p(ws), ws.numel() * ws.element_size(), self._stream()), 4.0 * y.numel() * 5),
            "sgg_layernorm_hwc_elu_bwd")

    # -- heads -----------------------------------------------------------------------------------------
    def spatial_mean_fwd(self, ctx, out_c, out_h):
        """ctx [B,L,C]; out_c / out_h: [R,C] (strided) views; row r gets mean_l ctx[r % B]."""
        self._dev(ctx, out_c, out_h)
        B, L, C = ctx.shape
        R = out_c.shape[0]
        self._check(self.lib.sgg_spatial_mean_fwd(_p(ctx), _p(out_c), _ld(out_c), _p(out_h), _ld(out_h), R, B, L, C,
                                                  self._stream()), "sgg_spatial_mean_fwd")

    def spatial_mean_bwd(self, dc0, dh0, dctx, accumulate):
        self._dev(dc0, dh0, dctx)
        B, L, C = dctx.shape
        R = dc0.shape[0]
        self._check(self.lib.sgg_spatial_mean_bwd(_p(dc0), _ld(dc0), _p(dh0), _ld(dh0), _p(dctx), R, B, L, C,
                                                  int(accumulate), self._stream()), "sgg_spatial_mean_bwd")

    def _gemm(self, mode, M, N, K, A, Bm, C, bias, accumulate):
        self._dev(A, Bm, C, bias)
        assert A.stride(1) == 1 and Bm.stride(1) == 1 and C.stride(1) == 1
        need = self.lib.sgg_gemm_workspace_bytes(M, N, K)
        ws = self.workspace(need)
        if mode == 0:
            rc = self.lib.sgg_gemm_skinny_fwd(M, N, K, _p(A), _ld(A), _p(Bm), _ld(Bm), _p(C), _ld(C), _p(bias),
                                              int(accumulate), _p(ws), ws.numel(), self._stream())
        else:
            fn = self.lib.sgg_gemm_skinny_dgrad if mode == 1 else self.lib.sgg_gemm_skinny_wgrad
            rc = fn(M, N, K, _p(A), _ld(A), _p(Bm), _ld(Bm), _p(C), _ld(C), int(accumulate), _p(ws), ws.numel(),
                    self._stream())
        self._check(rc, ("sgg_gemm_skinny_fwd", "sgg_gemm_skinny_dgrad", "sgg_gemm_skinny_wgrad")[mode])

    def gemm_nn(self, A, Bm, C, bias=None, accumulate=False):
        """C[M,N] (+)= A[M,K] @ B[K,N] (+ bias)"""
        M, K = A.shape
        K2, N = Bm.shape
        assert K == K2 and tuple(C.shape) == (M, N)
        self._gemm(0, M, N, K, A, Bm, C, bias, accumulate)

    def gemm_nt(self, A, Bm, C, accumulate=False):
        """C[M,N] (+)= A[M,K] @ B[N,K]^T"""
        M, K = A.shape
        N, K2 = Bm.shape
        assert K == K2 and tuple(C.shape) == (M, N)
        self._gemm(1, M, N, K, A, Bm, C, None, accumulate)

    def gemm_tn(self, A, Bm, C, accumulate=False):
        """C[M,N] (+)= A[K,M]^T @ B[K,N]"""
        K, M = A.shape
        K2, N = Bm.shape
        assert K == K2 and tuple(C.shape) == (M, N)
        self._gemm(2, M, N, K, A, Bm, C, None, accumulate)

    def attn_ctx_fwd(self, ctx_flat, W_ctx, bias, P):
        """P[B,L] = ctx_flat[B,L*C] @ W_ctx[L*C,L] + bias: the step-invariant part of the attention perceptron."""
        self._dev(ctx_flat, W_ctx, bias, P)
        B, LC = ctx_flat.shape
        L = P.shape[1]
        assert ctx_flat.is_contiguous() and W_ctx.is_contiguous() and P.is_contiguous() and tuple(W_ctx.shape) == (LC, L)
        ws = self.workspace(self.lib.sgg_gemm_workspace_bytes(B, L, LC))
        nb = 4.0 * (LC * L + B * LC + B * L)       # W_ctx streamed once + ctx + P
        self._check(self._timed("attn_ctx_gemm_fwd(call: split-K gemm + reduce)", 2.0 * B * L * LC, lambda: self.lib.sgg_attn_ctx_gemm_fwd(
            B, L, LC, _p(ctx_flat), _p(W_ctx), _p(bias), _p(P), _p(ws), ws.numel(), self._stream()), nb), "sgg_attn_ctx_gemm_fwd")

    def attn_ctx_dgrad(self, dP, W_ctx, dctx_flat, accumulate=True):
        self._dev(dP, W_ctx, dctx_flat)
        B, L = dP.shape
        LC = W_ctx.shape[0]
        assert dP.is_contiguous() and W_ctx.is_contiguous() and dctx_flat.is_contiguous()
        ws = self.workspace(self.lib.sgg_gemm_workspace_bytes(B, LC, L))
        nb = 4.0 * (LC * L + B * L + B * LC * (2 if accumulate else 1))
        self._check(self._timed("attn_ctx_gemm_dgrad(call)", 2.0 * B * L * LC, lambda: self.lib.sgg_attn_ctx_gemm_dgrad(
            B, L, LC, _p(dP), _p(W_ctx), _p(dctx_flat), int(accumulate), _p(ws), ws.numel(), self._stream()), nb),
            "sgg_attn_ctx_gemm_dgrad")

    def attn_ctx_wgrad(self, ctx_flat, dP, dW_ctx, accumulate=True):
        self._dev(ctx_flat, dP, dW_ctx)
        B, LC = ctx_flat.shape
        L = dP.shape[1]
        assert ctx_flat.is_contiguous() and dP.is_contiguous() and dW_ctx.is_contiguous()
        ws = self.workspace(self.lib.sgg_gemm_workspace_bytes(LC, L, B))
        nb = 4.0 * (LC * L * (2 if accumulate else 1) + B * L + B * LC)
        self._check(self._timed("attn_ctx_gemm_wgrad(call)", 2.0 * B * L * LC, lambda: self.lib.sgg_attn_ctx_gemm_wgrad(
            B, L, LC, _p(ctx_flat), _p(dP), _p(dW_ctx), int(accumulate), _p(ws), ws.numel(), self._stream()), nb),
            "sgg_attn_ctx_gemm_wgrad")

    @staticmethod
    def _planes(t):
        """[np, R, W] (strided) -> (ptr_real, ptr_dual or None, ld)"""
        if t is None:
            return None, None, 0
        assert t.dim() == 3 and t.stride(2) == 1
        ld = t.stride(1) if t.shape[1] > 1 else max(t.stride(1), t.shape[2])
        return t[0].data_ptr(), (t[1].data_ptr() if t.shape[0] == 2 else None), ld

    def attn_step_fwd(self, P, ec, ctx, alpha, z):
        """P [B,L]; ec, alpha [np,R,L]; z [np,R,C] view; ctx [B,L,C]."""
        self._dev(P, ec, ctx, alpha, z)
        B, L, C = ctx.shape
        R = ec.shape[1]
        er, ed, lde = self._planes(ec)
        ar, ad, lda = self._planes(alpha)
        zr, zd, ldz = self._planes(z)
        assert lda == L
        np_ = ec.shape[0]
        nb = 4.0 * (B * L * C + B * L + np_ * R * (2 * L + C))     # feature map once per image + P + ec, alpha, z
        self._check(self._timed("attn_step_fwd_kernel<%s>" % ("Dual" if np_ == 2 else "float"), 0.0, lambda: self.lib.sgg_attn_step_fwd(
            _p(P), er, ed, lde, _p(ctx), ar, ad, zr, zd, ldz, R, B, L, C, self._stream()), nb), "sgg_attn_step_fwd")

    def attn_step_bwd(self, ctx, alpha, dz, de, dP, dctx, accumulate):
        self._dev(ctx, alpha, dz, de, dP, dctx)
        B, L, C = ctx.shape
        R = alpha.shape[1]
        ar, ad, lda = self._planes(alpha)
        zr, zd, ldz = self._planes(dz)
        er, ed, lde = self._planes(de)
        assert lda == L and lde == L
        np_ = alpha.shape[0]
        nb = 4.0 * (B * L * C * (3 if accumulate else 2) + B * L * 2 + np_ * R * (2 * L + C))   # ctx read, dctx read+write
        self._check(self._timed("attn_step_bwd(call)<%s>" % ("Dual" if np_ == 2 else "float"), 0.0, lambda: self.lib.sgg_attn_step_bwd(
            _p(ctx), ar, ad, zr, zd, ldz, er, ed, _p(dP), _p(dctx), R, B, L, C, int(accumulate), self._stream()), nb),
            "sgg_attn_step_bwd")

    def lstm_fwd(self, gates, c_prev, ln_params, c_new, h_new):
        """gates [np,R,2048], c_prev/c_new [np,R,512] contiguous, h_new [np,R,512] view, ln_params [10,512]."""
        self._dev(gates, c_prev, ln_params, c_new, h_new)
        R = gates.shape[1]
        gr, gd, ldg = self._planes(gates)
        cr, cd, ldc = self._planes(c_prev)
        nr, nd, ldn = self._planes(c_new)
        hr, hd, ldh = self._planes(h_new)
        assert ldg == 2048 and ldc == 512 and ldn == 512
        np_ = gates.shape[0]
        nb = 4.0 * (np_ * R * (2048 + 512 + 1024) + 10 * 512)     # gates + c in, c', h' out (+ the LN vectors)
        self._check(self._timed("lnlstm_gates_fwd_kernel<%s>" % ("Dual" if np_ == 2 else "float"), 0.0, lambda: self.lib.sgg_lnlstm_gates_fwd(
            gr, gd, cr, cd, _p(ln_params), nr, nd, hr, hd, ldh, R, self._stream()), nb), "sgg_lnlstm_gates_fwd")

    def lstm_bwd(self, gates, c_prev, ln_params, dh, dc_new, dgates, dc_prev, pgrad):
        """dh [np,R,512] view; dc_new [np,R,512] or None; outputs dgates [np,R,2048], dc_prev [np,R,512], pgrad [R,10,512]."""
        self._dev(gates, c_prev, ln_params, dh, dc_new, dgates, dc_prev, pgrad)
        R = gates.shape[1]
        gr, gd, ldg = self._planes(gates)
        cr, cd, ldc = self._planes(c_prev)
        hr, hd, ldh = self._planes(dh)
        nr, nd, ldn = self._planes(dc_new)
        dgr, dgd, lddg = self._planes(dgates)
        dpr, dpd, lddp = self._planes(dc_prev)
        assert ldg == 2048 and ldc == 512 and lddg == 2048 and lddp == 512 and (dc_new is None or ldn == 512)
        np_ = gates.shape[0]
        nb = 4.0 * (np_ * R * (2048 + 512 + 512 + (512 if dc_new is not None else 0) + 2048 + 512) + R * 5120 + 5120)
        self._check(self._timed("lnlstm_gates_bwd_kernel<%s>" % ("Dual" if np_ == 2 else "float"), 0.0, lambda: self.lib.sgg_lnlstm_gates_bwd(
            gr, gd, cr, cd, _p(ln_params), hr, hd, ldh, nr, nd, dgr, dgd, dpr, dpd, _p(pgrad), R, self._stream()), nb),
            "sgg_lnlstm_gates_bwd")

    def colsum(self, X, out, accumulate=False):
        self._dev(X, out)
        rows, cols = X.shape
        assert X.stride(1) == 1
        ws = self.workspace(self.lib.sgg_colsum_workspace_bytes(rows, cols))
        self._check(self.lib.sgg_colsum(_p(X), rows, cols, _ld(X), _p(out), int(accumulate), _p(ws), ws.numel(), self._stream()),
                    "sgg_colsum")

    # -- loss / optimiser / misc -----------------------------------------------------------------------
    def onehot(self, labels, out):
        self._dev(labels, out)
        assert labels.dtype == torch.int64
        V = out.shape[-1]
        self._check(self.lib.sgg_onehot(_p(labels), _p(out), labels.numel(), V, self._stream()), "sgg_onehot")

    def embed_gather_fwd(self, labels, W, out):
        """labels: int64 [R] (strided) view; W [V,E]; out [R,E] view: out[r] = W[labels[r]] (tf.matmul(one_hot, W))."""
        self._dev(labels, W, out)
        assert labels.dtype == torch.int64 and labels.dim() == 1 and W.is_contiguous() and out.stride(1) == 1
        V, E = W.shape
        R = labels.shape[0]
        self._check(self.lib.sgg_embed_gather_fwd(_p(labels), max(labels.stride(0), 1), _p(W), V, E, _p(out), _ld(out), R,
                                                  self._stream()), "sgg_embed_gather_fwd")

    def embed_gather_bwd(self, labels, dY, dW):
        """dW[labels[r]] += dY[r] (deterministic)."""
        self._dev(labels, dY, dW)
        assert labels.dtype == torch.int64 and labels.dim() == 1 and dW.is_contiguous() and dY.stride(1) == 1
        V, E = dW.shape
        R = labels.shape[0]
        self._check(self.lib.sgg_embed_gather_bwd(_p(labels), max(labels.stride(0), 1), _p(dY), _ld(dY), _p(dW), V, E, R,
                                                  self._stream()), "sgg_embed_gather_bwd")

    def resize_bilinear_tf1(self, packed, offsets, heights, widths, out, means, stds):
        """packed uint8 [nbytes] (RGB images back to back), offsets int64 [B], heights / widths int32 [B] -> out [B,oh,ow,3] fp32 =
        (tf.image.resize_images(img, [oh, ow]) - means) / stds  (train.py:171-172)."""
        self._dev(packed, offsets, heights, widths, out, means, stds)
        assert packed.dtype == torch.uint8 and offsets.dtype == torch.int64 and heights.dtype == torch.int32 and widths.dtype == torch.int32
        B, oh, ow, _ = out.shape
        self._check(self.lib.sgg_resize_bilinear_tf1(_p(packed), _p(offsets), _p(heights), _p(widths), _p(out), B, oh, ow, _p(means),
                                                     _p(stds), self._stream()), "sgg_resize_bilinear_tf1")

    def interpolate(self, real, fake, alpha, out):
        self._dev(real, fake, alpha, out)
        B = real.shape[0]
        self._check(self.lib.sgg_interpolate(_p(real), _p(fake), _p(alpha), _p(out), B, real.numel() // B, self._stream()),
                    "sgg_interpolate")

    def gp_fwd(self, g, slopes, pen):
        self._dev(g, slopes, pen)
        B = g.shape[0]
        self._check(self.lib.sgg_wgan_gp_loss_fwd(_p(g), _p(slopes), _p(pen), B, g.numel() // B, self._stream()),
                    "sgg_wgan_gp_loss_fwd")

    def gp_bwd(self, g, slopes, pen, v, scale):
        self._dev(g, slopes, pen, v)
        B = g.shape[0]
        self._check(self.lib.sgg_wgan_gp_loss_bwd(_p(g), _p(slopes), _p(pen), _p(v), B, g.numel() // B, float(scale),
                                                  self._stream()), "sgg_wgan_gp_loss_bwd")

    def wgan_losses(self, d_out, pen, lam, B, T, has_real, out4):
        self._dev(d_out, pen, out4)
        self._check(self.lib.sgg_wgan_losses(_p(d_out), _p(pen), float(lam), B, T, int(has_real), _p(out4), self._stream()),
                    "sgg_wgan_losses")

    def adam(self, params, grads, m, v, lr_t, b1, b2, eps, grad_scale=1.0):
        self._dev(params, grads, m, v)
        self._check(self._timed("adam_kernel", 0.0, lambda: self.lib.sgg_adam_tf_multi(
            _p(params), _p(grads), _p(m), _p(v), params.numel(), float(lr_t), float(b1), float(b2), float(eps), float(grad_scale),
            self._stream()), 28.0 * params.numel()), "sgg_adam_tf_multi")

    def argmax_rows(self, x, out):
        self._dev(x, out)
        assert x.stride(-1) == 1 and out.dtype == torch.int64
        V = x.shape[-1]
        x2 = x.reshape(-1, V)
        self._check(self.lib.sgg_argmax_rows(_p(x2), _p(out), x2.shape[0], V, _ld(x2), self._stream()), "sgg_argmax_rows")

    def fill(self, t, value):
        self._dev(t)
        assert t.is_contiguous()
        self._check(self.lib.sgg_fill(_p(t), t.numel(), float(value), self._stream()), "sgg_fill")
