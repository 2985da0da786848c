"""Convolutional encoder (12 live conv layers, 11 of them followed by LayerNorm(H,W,C)+ELU): forward and the
hand-scheduled backward over the HIP kernels.

Reference: architectures/generator_with_attention.py:29-68 == architectures/discriminator_with_attention.py:29-68
(forward); gradients as produced by optimizer.minimize (train.py:265-266).  conv3_3/conv3_4 and their
LayerNorms (:59-62) never reach `downsampled` and are neither computed nor given gradients.

HBM layout: per layer the conv output y_i and the activation a_i = ELU(LN(y_i)) stay resident for the backward
pass (5.6 GB per network at batch 64 / 224x224 - sized for 288 GB of HBM3E, nothing is recomputed); the
backward pass ping-pongs two scratch tensors (dA, dY) of the largest activation size.

Odd image sizes (the reference trains on 221 x 221, train.py:171: maps of 221, 111, 56, 28, 14) run on EVEN CANVASES
(224, 112, 56, 28, 14) so that the tiled halo / band / halo-wgrad kernels apply: the true map of a stage sits at an offset inside
its canvas (3 for 221, 1 for 111, 0 from 56 on), chosen such that the canvas convolution's own SAME padding lines up with the true
one; the LayerNorm kernels normalise the valid window only and write zeros outside it, which is exactly the zero padding the
next convolution (and, in the backward pass, the next dgrad / wgrad) must see.  See `plan_canvas`.
"""
from __future__ import annotations

import torch

from .params import CONV_SPECS, FEAT_C, conv_name, ln_name, same_pads


def plan_canvas(S):
    """Per live conv layer: (canvas_in, offset_in, true_in, canvas_out, offset_out, true_out), or None when the plain grid is used.

    A canvas convolution (SAME padding computed from the canvas size, pad_c in front) reproduces the true convolution (pad_t in
    front) at out[q] = true_out[q - o_out] iff  o_in = stride * o_out + pad_t - pad_c  and the input is zero outside its valid
    window.  Working back from the last layer (offset 0, canvas = true size) this fixes every offset; the plan is rejected when an
    offset would be negative, a window would not fit its canvas or the canvas is over 1.3x the image (then the plain, possibly odd,
    grid is used)."""
    specs = [(k, s) for (_, _, _, k, s, _, live) in CONV_SPECS if live]
    true = [S]
    for k, s in specs:
        true.append(same_pads(true[-1], k, s)[0])
    canvas, off = [0] * len(true), [0] * len(true)
    canvas[-1] = true[-1]
    for j in range(len(specs) - 1, -1, -1):
        k, s = specs[j]
        canvas[j] = canvas[j + 1] * s
        pad_t, pad_c = same_pads(true[j], k, s)[1], same_pads(canvas[j], k, s)[1]
        off[j] = s * off[j + 1] + pad_t - pad_c
        if off[j] < 0 or off[j] + true[j] > canvas[j] or same_pads(canvas[j], k, s)[0] != canvas[j + 1]:
            return None
    if canvas == true or canvas[0] ** 2 > 1.3 * true[0] ** 2:      # (nothing to do / the canvas would cost more than it saves)
        return None
    return [(canvas[j], off[j], true[j], canvas[j + 1], off[j + 1], true[j + 1]) for j in range(len(specs))]


def ln_fusion_pays(out_shape, cout_next):
    """(forward-only pass, pass followed by a backward): does applying this layer's LayerNorm + ELU in the consumer's patch staging
    beat the LayerNorm apply pass?  Measured cost model at batch 64 (DESIGN.md "The LN prologue"), microseconds, MB = the
    activation's bytes / 1e6:
      saved      0.35 * MB                               read y + write a at 5.7 TB/s
      forward    0.09 * MB * max(1, Cout / 128)          consumer on a two-block variant (128- / 64-column tiles; x staged once per 128
                                                         output columns)
                 55                                      consumer on the four-block variant (32-column tiles): flat
      wgrad      0.075 * MB * Cout / 64                  (x staged once per 64 output columns), only when a backward follows; with a
                                                         64-column consumer the pass with backward measured no gain (47.77 / 47.86 vs
                                                         47.71 / 47.82 ms per step): forward-only there"""
    mb = 4e-6
    for d in out_shape:
        mb *= d
    saved = 0.35 * mb
    fwd_cost = 0.09 * mb * max(1.0, cout_next / 128.0) if cout_next % 64 == 0 else 55.0
    wgrad_cost = 0.075 * mb * cout_next / 64.0
    return saved > 1.5 * fwd_cost, cout_next != 64 and saved - fwd_cost - wgrad_cost > 10.0


def pc_ln_fusion_pays(out_shape, cin_next, cout_next):
    """Forward-only passes, consumer on the producer / consumer 3x3 kernel (which otherwise stages a pre-split a_j by LDS-DMA): does the
    LN prologue - the consumer's register-staging variant with the prologue in its producer waves, ~340 against ~425 TFLOP/s - beat the
    LayerNorm apply pass it saves (0.35 us per MB, ln_fusion_pays)?  Round 5, batch 64, two boxes (profiles/r05_ln_plan_ab*.log):
    LN5 (411 MB in front of conv2_4) -0.19 ms per step, LN4 (205 MB in front of conv2_3) -0.09, both -0.2 ... -0.25; LN7 / LN8 (103 /
    205 MB in front of conv3_1 / conv3_2, the same FLOPs behind a quarter / half of the bytes): none - which is what the two rates
    predict (72 vs 70 us, 144 vs 139; 36 vs 70, 72 vs 139)."""
    b, h, w, c = out_shape
    mb = 4e-6 * b * h * w * c
    flops = 2.0 * b * h * w * cout_next * 9 * cin_next
    return 0.35 * mb > flops * (1.0 / 340e12 - 1.0 / 425e12) * 1e6


class Trunk:
    def __init__(self, K, arena, grad_views, B, S):
        self.K, self.B, self.S = K, B, S
        self.arena = arena
        dev, dt = arena.flat.device, arena.flat.dtype
        p, g = arena.views, grad_views
        self.layers = []
        plan = plan_canvas(S) if getattr(K, "canvas", True) else None
        S_in = plan[0][0] if plan else S
        # the image inside its zero canvas (odd sizes): filled by forward(), zero elsewhere for good
        self.img_canvas = torch.zeros((B, S_in, S_in, 3), device=dev, dtype=dt) if plan else None
        self.img_off = plan[0][1] if plan else 0
        h = w = S_in
        cin_shape = (B, S_in, S_in, 3)
        max_act = 0
        li = -1
        for (i, cin, cout, k, s, has_ln, live) in CONV_SPECS:
            if not live:
                continue
            li += 1
            ho, wo = same_pads(h, k, s)[0], same_pads(w, k, s)[0]
            region = None
            if plan:
                c_out, o_out, t_out = plan[li][3:]
                assert c_out == ho
                if (c_out, o_out) != (t_out, 0):
                    region = (o_out, o_out, t_out, t_out)      # valid window of this layer's output canvas
            lay = {
                "region": region,
                "i": i, "cin": cin, "cout": cout, "k": k, "s": s, "has_ln": has_ln,
                "in_shape": cin_shape, "out_shape": (B, ho, wo, cout),
                "w": p[conv_name(i) + "/kernel"], "b": p[conv_name(i) + "/bias"],
                "gw": g[conv_name(i) + "/kernel"], "gb": g[conv_name(i) + "/bias"],
                "y": torch.empty((B, ho, wo, cout), device=dev, dtype=dt),
            }
            lay["w_fwd"] = lay["w"] if cin == 3 else torch.empty((k, k, cout, cin), device=dev, dtype=dt)
            # split modes: both weight layouts pre-split into 16-bit planes once per optimiser step (HIP backend only)
            lay["ws_fwd"] = lay["ws_bwd"] = None
            # 3x3 stride-1 layers on 8-divisible grids: halo-resident kernel, weights pre-arranged as MFMA fragments
            # (re-derived in refresh_weights: the layout depends on the conv precision in force)
            lay["hin"], lay["win"] = h, w
            lay["prev"] = self.layers[-1] if self.layers else None
            # forward and dgrad are asked separately: the kernels' channel conditions are not symmetric in (cin, cout)
            lay["ws_layout"] = lay["ws_layout_bwd"] = 0
            self._query_layouts(lay)
            if cin != 3 and getattr(K, "conv_precision", 0) and hasattr(K, "split_weights"):
                lay["ws_fwd"] = torch.empty((3, lay["w"].numel()), device=dev, dtype=torch.int16)
                lay["ws_bwd"] = torch.empty((3, lay["w"].numel()), device=dev, dtype=torch.int16)
            if 3 in (lay["ws_layout"], lay["ws_layout_bwd"]):
                # conv1_3 as a 3x3 convolution over the space-to-depth view of its input: the 9-tap kernel and its HWOI transpose
                lay["w3"] = torch.empty((3, 3, 4 * cin, cout), device=dev, dtype=dt)
                lay["w3_fwd"] = torch.empty((3, 3, cout, 4 * cin), device=dev, dtype=dt)
            lay["tstats"] = None
            if has_ln and hasattr(K, "conv_tile_stats_count") and region is None:
                nts = K.conv_tile_stats_count((B, ho, wo, cout), cin, k, s, lay["ws_layout"])
                if nts > 0:     # the conv epilogue emits the LayerNorm partial statistics for this shape
                    lay["tstats"] = torch.zeros((B, nts, 4), device=dev, dtype=dt)
                    lay["tstats_mode"] = (K.conv_precision, lay["ws_layout"])
            if has_ln:
                lay["gamma"], lay["beta"] = p[ln_name(i) + "/gamma"], p[ln_name(i) + "/beta"]
                lay["ggamma"], lay["gbeta"] = g[ln_name(i) + "/gamma"], g[ln_name(i) + "/beta"]
                lay["a"] = torch.empty((B, ho, wo, cout), device=dev, dtype=dt)
                lay["stats"] = torch.empty((B, 2), device=dev, dtype=dt)
                max_act = max(max_act, B * ho * wo * cout)
            self.layers.append(lay)
            h, w, cin_shape = ho, wo, (B, ho, wo, cout)
        self.Hf, self.Wf = h, w
        self.L = h * w
        self._dA = torch.empty(max_act, device=dev, dtype=dt)
        self._dY = torch.empty(max_act, device=dev, dtype=dt)
        # f16x3 mode: device words with max|tensor| of every conv operand (rows: activations a_j, gradients dy_j,
        # weights w_j), maintained by the producing kernels, so operands can be scaled into fp16 range without a host sync
        self.amax = torch.zeros((3, 16), device=dev, dtype=torch.float32)
        # pre-split dy (LayerNorm backward): per layer the two maxima its bound is made of (zeroed with one fill per backward)
        self.pq = torch.zeros((16, 2), device=dev, dtype=torch.float32)
        # LayerNorm backward: the parameter-gradient reductions of all layers in ONE launch at the end of the encoder backward
        # (K.ln_bwd_finalize), from per-layer workspaces that keep the partial sums until then
        self._ln_fin = None
        if hasattr(K, "ln_bwd_finalize"):
            todo = []
            for lay in self.layers:
                if lay["has_ln"]:
                    lay["ln_ws"] = torch.empty(K.ln_workspace_bytes(lay["out_shape"]), device=dev, dtype=torch.uint8)
                    todo.append({"ws": lay["ln_ws"], "gamma": lay["gamma"], "stats": lay["stats"], "dgamma": lay["ggamma"], "dbeta": lay["gbeta"],
                                 "dbias": lay["gb"], "shape": lay["out_shape"], "region": lay["region"]})
            self._ln_fin = K.ln_finalize_descs(todo)
        self.refresh_weights()

    def _query_layouts(self, lay):
        K = self.K
        if lay["cin"] != 3 and hasattr(K, "conv_wsplit_layout"):
            k, s = lay["k"], lay["s"]
            lay["ws_layout"] = K.conv_wsplit_layout(k, s, lay["hin"], lay["win"], lay["cin"], lay["cout"])
            if lay["ws_layout"] == 4 and self._ln_prologue_expected(lay):
                # The producer / consumer 3x3 kernel (layout 4) wins without the LayerNorm prologue (408-413 against 384 TFLOP/s in the
                # step) and loses with it (337-345 against 358: its four producer waves carry all of the prologue's v_exp work beside one
                # MFMA wave per SIMD).  A layer whose forward mostly runs with the prologue keeps the four-wave kernel's fragments for
                # the forward; its dgrad (never a prologue) still takes layout 4 below.  
                lay["ws_layout"] = 1
            ho, wo = lay["out_shape"][1], lay["out_shape"][2]
            # (both directions are asked with the full-resolution grid: forward input = dgrad output)
            lay["ws_layout_bwd"] = K.conv_wsplit_layout(k, s, lay["hin"], lay["win"], lay["cout"], lay["cin"])
            if lay["ws_layout_bwd"] == 1 and self._dy_presplit_static(lay) and hasattr(K, "conv_wsplit_layout_presplit"):
                # the dgrad of a 64-column layer whose dy the LayerNorm backward writes pre-split (conv2_2, conv2_3): the four-block
                # form of the producer / consumer kernel (patch by LDS-DMA only: the launch always passes dy_s16, backward())
                lay["ws_layout_bwd"] = K.conv_wsplit_layout_presplit(k, s, lay["hin"], lay["win"], lay["cout"], lay["cin"])

    def _dy_presplit_static(self, lay):
        """Will _plan_s16 / backward() hand this layer's dgrad a PRE-SPLIT dy in every backward?  (The static conditions of
        lay["dy_s16"] and of backward()'s nxt_s16: LayerNorm layer, fp16 two-piece mode with pre-split weights, the deferred LayerNorm
        finalize, a resident filter gradient.)"""
        K = self.K
        if not (lay["has_ln"] and lay["cin"] != 3 and bool(getattr(K, "presplit", False)) and getattr(K, "conv_precision", 0) == 2
                and hasattr(K, "wgrad_resident") and hasattr(K, "ln_bwd_finalize") and hasattr(K, "split_weights")):
            return False
        Bn, ho, wo, cout = lay["out_shape"]
        return bool(K.wgrad_resident(Bn, ho, wo, lay["cin"], lay["cout"], lay["k"], lay["s"]))

    def _ln_prologue_expected(self, lay):
        """Will _plan_ln_fusion let this layer's forward apply the previous layer's LayerNorm + ELU in its patch staging, in either
        kind of pass?  (The static part of that plan: the statistics conditions it adds can only remove a fusion.)"""
        K, prev = self.K, lay.get("prev")
        mode = getattr(K, "ln_fusion", 0)
        if not mode or prev is None or not prev["has_ln"] or prev["region"] or not hasattr(K, "ln_prologue_ok"):
            return False
        if prev["i"] in getattr(K, "ln_fusion_skip", ()):
            return False
        if mode == 1 and self._pc_presplit(lay):
            return False
        return mode == 2 or any(ln_fusion_pays(prev["out_shape"], lay["cout"]))

    def _pc_presplit(self, lay):
        """With pre-split activations a consumer on the producer / consumer 3x3 kernel (128-column layers) stages its patch by
        LDS-DMA and its filter gradient runs on the LDS-DMA kernel; an LN prologue would put both back on the register-staging
        kernels.  Measured with the prologue kept for these consumers: 45.29 / 45.20 against 45.09 / 45.16 ms per step without
        (profiles/r04_ln_fusion_plan_presplit_ab.log) - and the plain producer / consumer kernel is the faster kernel (0.50 of the
        matrix peak against 0.43): those LayerNorms keep their apply pass - in passes with a backward always; in forward-only passes
        except where pc_ln_fusion_pays says the saved pass is worth the slower variant (round 5: LN4, LN5)."""
        K = self.K
        return (bool(getattr(K, "presplit", False)) and getattr(K, "conv_precision", 0) == 2 and hasattr(K, "conv_wsplit_layout") and
                K.conv_wsplit_layout(lay["k"], lay["s"], lay["hin"], lay["win"], lay["cin"], lay["cout"]) == 4)

    def _f16(self):
        return getattr(self.K, "conv_precision", 0) in (1, 2)      # fp16 pieces: per-tensor scaling from the amax words

    def _am(self, row, j):
        return self.amax[row, j:j + 1] if self._f16() else None

    def _plan_ln_fusion(self):
        """lay["fuse_ln"] / lay["fuse_ln_bwd"]: this layer's LayerNorm + ELU is applied by its consumer (the next convolution's patch
        staging) instead of a pass of its own, in encoder passes that no backward follows / that a backward follows (then the
        activation a_j is never written and the consumer's wgrad applies the prologue as well).  Needs the statistics partials from
        this layer's conv epilogue and a consumer on the halo-resident kernels in the conv precision in force.

        K.ln_fusion: 0 off; 2 wherever the kernels allow (slower: DESIGN.md); 1 (default) where the measured cost model
        (ln_fusion_pays) says it pays, per kind of pass."""
        K = self.K
        mode = getattr(K, "ln_fusion", 0)
        for j, lay in enumerate(self.layers):
            lay["fuse_ln"] = lay["fuse_ln_bwd"] = False
            if not (lay["has_ln"] and hasattr(K, "ln_prologue_ok") and mode) or j + 1 >= len(self.layers) or lay["region"]:
                continue
            nxt = self.layers[j + 1]
            stats_ok = lay["tstats"] is not None and (lay["cin"] == 3 or (lay["ws_fwd"] is not None and lay.get("ws_mode") == K.conv_precision
                                                                          and lay.get("tstats_mode") == (K.conv_precision, lay["ws_layout"])))
            if not (stats_ok and nxt["ws_fwd"] is not None):
                continue
            args = (nxt["k"], nxt["s"], nxt["hin"], nxt["win"], nxt["cin"], nxt["cout"])
            both_ok = bool(K.ln_prologue_ok(*args))
            fwd_ok = both_ok or bool(hasattr(K, "ln_prologue_fwd_ok") and K.ln_prologue_fwd_ok(*args))
            if mode == 2:
                lay["fuse_ln"] = lay["fuse_ln_bwd"] = both_ok
                continue
            pays_fwd, pays_bwd = ln_fusion_pays(lay["out_shape"], nxt["cout"])
            if self._pc_presplit(nxt):
                # consumer on the producer / consumer kernel: with a backward never (its filter gradient wants the pre-split a_j);
                # forward-only where the apply pass it saves outweighs the consumer's slower prologue variant (pc_ln_fusion_pays)
                pays_fwd, pays_bwd = pc_ln_fusion_pays(lay["out_shape"], nxt["cin"], nxt["cout"]), False
            if lay["i"] in getattr(K, "ln_fusion_skip", ()):      # (A/B option ln_fusion_skip of sgg_amd/lib.py)
                pays_fwd = pays_bwd = False
            if lay["i"] in getattr(K, "ln_fusion_force", ()):     # (A/B option ln_fusion_force)
                pays_fwd = True
            if lay["i"] in getattr(K, "ln_fusion_force_bwd", ()):
                pays_bwd = True
            if lay["i"] in getattr(K, "ln_fusion_skip_bwd", ()):  # (A/B option: keep the apply pass in passes a backward follows)
                pays_bwd = False
            lay["fuse_ln"] = fwd_ok and pays_fwd
            lay["fuse_ln_bwd"] = both_ok and pays_bwd

    def _plan_s16(self):
        """lay["a_s16"] / lay["dy_s16"]: this layer's activation a_j = ELU(LN(y_j)) / the gradient dy_j its LayerNorm backward produces
        is written PRE-SPLIT (two fp16 pieces per value in the f32 tensor's bytes: csrc/split16.h) by the LayerNorm kernel, so that the
        convolutions that consume it stage it without splitting it again.  Needs the fp16 modes (per-tensor scale) and every consumer on
        a resident kernel: a_j feeds conv_{j+1}'s forward and filter gradient, dy_j feeds conv_j's dgrad and filter gradient."""
        K = self.K
        on = bool(getattr(K, "presplit", False)) and getattr(K, "conv_precision", 0) in (1, 2) and hasattr(K, "wgrad_resident")
        n = len(self.layers)
        for j, lay in enumerate(self.layers):
            lay["a_s16"] = lay["dy_s16"] = False
            if not on:
                continue
            if not lay["has_ln"]:
                # the last convolution: its dy is the head's f32 gradient, converted once per backward (sgg_presplit16) where both of
                # its consumers take pre-split operands and its input activation arrives pre-split as well
                wg = K.wgrad_resident(lay["out_shape"][0], lay["out_shape"][1], lay["out_shape"][2], lay["cin"], lay["cout"], lay["k"], lay["s"])
                lay["dy_s16"] = (j == n - 1 and j > 0 and hasattr(K, "presplit16") and bool(getattr(K, "presplit_head_grad", True)) and
                                 lay["ws_bwd"] is not None and
                                 lay["ws_layout_bwd"] in (1, 2, 3, 4) and wg and lay["cout"] % 32 == 0)
                continue
            B, ho, wo, cout = lay["out_shape"]
            wg_ok = lambda l: K.wgrad_resident(B, l["out_shape"][1], l["out_shape"][2], l["cin"], l["cout"], l["k"], l["s"])
            if j + 1 < n:
                nxt = self.layers[j + 1]
                lay["a_s16"] = nxt["ws_fwd"] is not None and nxt["ws_layout"] in (1, 2, 3, 4) and wg_ok(nxt)
            lay["dy_s16"] = lay["cin"] != 3 and lay["ws_bwd"] is not None and lay["ws_layout_bwd"] in (1, 2, 3, 4) and wg_ok(lay)

    def refresh_weights(self):
        """Re-derive the operand formats of the convolution kernels after the parameters changed (Adam step / state-dict load):
        HWOI transposes, max|w| words, the space-to-depth kernel of conv1_3, both pre-split 16-bit copies.  One multi-tensor call
        (three launches for the whole encoder) where the kernel set has it; per layer otherwise (the CPU reference kernels)."""
        K = self.K
        self._wver = getattr(self.arena, "version", 0)
        if hasattr(K, "prepare_weights"):
            key = (K.conv_precision, getattr(K, "conv_halo", True), bool(getattr(K, "presplit", False)), bool(getattr(K, "halo_pc64", True)))
            if getattr(self, "_wdesc_key", None) != key:       # layouts depend on the precision in force: rebuild the table
                todo = []
                for j, lay in enumerate(self.layers):
                    if lay["cin"] == 3:
                        continue
                    self._query_layouts(lay)
                    if 3 in (lay["ws_layout"], lay["ws_layout_bwd"]) and "w3" not in lay:
                        lay["w3"] = torch.empty((3, 3, 4 * lay["cin"], lay["cout"]), device=lay["w"].device, dtype=lay["w"].dtype)
                        lay["w3_fwd"] = torch.empty((3, 3, lay["cout"], 4 * lay["cin"]), device=lay["w"].device, dtype=lay["w"].dtype)
                    split = lay["ws_fwd"] is not None and K.conv_precision
                    todo.append({"w": lay["w"], "w_fwd": lay["w_fwd"], "w3": lay.get("w3"), "w3_fwd": lay.get("w3_fwd"),
                                 "ws_fwd": lay["ws_fwd"] if split else None, "ws_bwd": lay["ws_bwd"] if split else None,
                                 "amax": self._am(2, j), "ws_layout": lay["ws_layout"] if split else 0,
                                 "ws_layout_bwd": lay["ws_layout_bwd"] if split else 0})
                self._wdesc, self._wdesc_key = K.weight_descs(todo), key
            K.prepare_weights(self._wdesc)
            for lay in self.layers:
                if lay["cin"] != 3 and lay["ws_fwd"] is not None and K.conv_precision:
                    lay["ws_mode"] = K.conv_precision
            self._plan_ln_fusion()
            self._plan_s16()
            return
        if self._f16():
            self.K.fill(self.amax[2], 0.0)
        for j, lay in enumerate(self.layers):
            if lay["cin"] != 3:
                self.K.hwio_to_hwoi(lay["w"], lay["w_fwd"])
                if self._f16():
                    self.K.absmax(lay["w"], self._am(2, j))
                if lay["ws_fwd"] is not None and self.K.conv_precision:
                    self._query_layouts(lay)
                    if 3 in (lay["ws_layout"], lay["ws_layout_bwd"]):
                        if "w3" not in lay:
                            lay["w3"] = torch.empty((3, 3, 4 * lay["cin"], lay["cout"]), device=lay["w"].device, dtype=lay["w"].dtype)
                            lay["w3_fwd"] = torch.empty((3, 3, lay["cout"], 4 * lay["cin"]), device=lay["w"].device, dtype=lay["w"].dtype)
                        self.K.s2d_weights(lay["w"], lay["w3"])
                        self.K.hwio_to_hwoi(lay["w3"], lay["w3_fwd"])
                    self.K.split_weights(lay["w3_fwd"] if lay["ws_layout"] == 3 else lay["w_fwd"], lay["ws_fwd"], self._am(2, j), lay["ws_layout"])
                    self.K.split_weights(lay["w3"] if lay["ws_layout_bwd"] == 3 else lay["w"], lay["ws_bwd"], self._am(2, j), lay["ws_layout_bwd"])
                    lay["ws_mode"] = self.K.conv_precision
        self._plan_ln_fusion()
        self._plan_s16()

    def forward(self, images, for_backward=True, cu_cap=0):
        """images [B,S,S,3] NHWC fp32, already standardised (train.py:172) -> downsampled as ctx [B, L, 512].
        for_backward=False (G in the critic update, D in the generator update: no encoder backward follows): LayerNorm + ELU of a
        layer may be applied by the consuming convolution's patch staging instead of a pass of its own (K.ln_fusion = 1).
        cu_cap (1 .. 31): launch hint - the persistent convolution kernels of this pass occupy at most that many of an XCD's 32 CUs
        (a forward that runs beside another stream's latency-critical chain: step.GanStep, option g_early_cus)."""
        cu_cap = cu_cap or int(getattr(self.K, "fwd_cus", 0))      # (option fwd_cus: the same hint for every encoder forward)
        cap = {"cu_cap": int(cu_cap)} if cu_cap else {}
        assert tuple(images.shape) == (self.B, self.S, self.S, 3), images.shape
        K = self.K
        if self._wver != getattr(self.arena, "version", 0):
            self.refresh_weights()          # the parameters changed through another encoder on this arena (another batch size)
        fuse_key = "fuse_ln_bwd" if for_backward else "fuse_ln"
        self._fwd_for_backward = for_backward
        if self.img_canvas is not None:
            o = self.img_off
            self.img_canvas[:, o:o + self.S, o:o + self.S].copy_(images)
            images = self.img_canvas
        self.images = images
        x = images
        x_s16 = False           # x is a pre-split activation (written so by the previous layer's LayerNorm)
        ln_in = None            # (stats, gamma, beta) when x is a pre-LayerNorm tensor whose LN + ELU this layer applies itself
        if self._f16():
            K.fill(self.amax[0], 0.0)
        for j, lay in enumerate(self.layers):
            ws = lay["ws_fwd"] if (lay["ws_fwd"] is not None and lay.get("ws_mode") == getattr(K, "conv_precision", 0)) else None
            ts = lay["tstats"] if (lay["tstats"] is not None and (ws is not None or lay["cin"] == 3)
                                   and (lay["cin"] == 3 or lay.get("tstats_mode") == (K.conv_precision, lay["ws_layout"]))) else None
            if ln_in is not None:
                K.conv_fwd(x, lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"], ws, self._am(0, j - 1), self._am(2, j), ts,
                           lay["ws_layout"], ln=ln_in, **cap)
            elif x_s16:
                assert ws is not None
                K.conv_fwd(x, lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"], ws, self._am(0, j - 1), self._am(2, j), ts, lay["ws_layout"],
                           x_s16=True, **cap)
            elif ws is not None or self._f16() or ts is not None:
                K.conv_fwd(x, lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"], ws, self._am(0, j - 1) if j else None, self._am(2, j), ts,
                           lay["ws_layout"] if ws is not None else 0, **cap)
            else:
                K.conv_fwd(x, lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"])
            ln_in = None
            x_s16 = False
            if lay["has_ln"]:
                if lay.get(fuse_key) and ts is not None:
                    # statistics only; the consumer normalises y while it stages its patches (forward and wgrad)
                    K.ln_finalize(ts, lay["gamma"], lay["beta"], lay["stats"], self._am(0, j), lay["out_shape"][1] * lay["out_shape"][2])
                    ln_in = (lay["stats"], lay["gamma"], lay["beta"])
                    x = lay["y"]
                    lay["fused_now"] = True
                    continue
                lay["fused_now"] = False
                # (the consumer must exist in the precision in force: its pre-split weights are what x_s16 above asserts)
                s16 = bool(lay.get("a_s16")) and j + 1 < len(self.layers) and self.layers[j + 1].get("ws_mode") == getattr(K, "conv_precision", 0)
                lay["a_s16_now"] = s16
                if s16:
                    K.ln_elu_fwd(lay["y"], lay["gamma"], lay["beta"], lay["a"], lay["stats"], self._am(0, j), ts, region=lay["region"], out_s16=True)
                elif lay["region"] is not None:
                    K.ln_elu_fwd(lay["y"], lay["gamma"], lay["beta"], lay["a"], lay["stats"], *([self._am(0, j), None] if self._f16() else []),
                                 region=lay["region"])
                elif self._f16() or ts is not None:
                    K.ln_elu_fwd(lay["y"], lay["gamma"], lay["beta"], lay["a"], lay["stats"], self._am(0, j), ts)
                else:
                    K.ln_elu_fwd(lay["y"], lay["gamma"], lay["beta"], lay["a"], lay["stats"])
                x = lay["a"]
                x_s16 = s16
            else:
                x = lay["y"]
        return x.view(self.B, self.L, FEAT_C)

    def enable_wgrad_overlap(self, stream):
        """Run the filter gradients on `stream` beside the dgrad -> LayerNorm-backward chain of the main stream (they only share
        their input dy_j): the MFMA-bound wgrad kernels fill the matrix cores while the HBM-bound LayerNorm passes run.  Costs a
        second dY scratch tensor (wgrad_j still reads dy_j while LayerNorm backward j-1 writes dy_{j-1})."""
        self.wgrad_stream = stream
        if stream is not None and getattr(self, "_dY2", None) is None:
            self._dY2 = torch.empty_like(self._dY)

    def backward(self, dctx):
        """dctx [B, L, 512]: gradient w.r.t. `downsampled`. Writes every live conv / LN parameter gradient."""
        K, B = self.K, self.B
        assert getattr(self, "_fwd_for_backward", True), "the last forward was run with for_backward=False"
        dy = dctx.view(B, self.Hf, self.Wf, FEAT_C)
        n = len(self.layers)
        f16 = self._f16()
        side = getattr(self, "wgrad_stream", None)
        main = torch.cuda.current_stream() if side is not None else None
        dybufs = [self._dY, self._dY2] if side is not None else [self._dY]
        reader_done = [None] * len(dybufs)        # event: the wgrad that reads this dY buffer has finished
        dy_s16 = False                             # dy is pre-split (written so by the LayerNorm backward of this layer)
        dy_f32 = dy                                # (the last convolution's bias gradient sums the f32 tensor)
        if f16:
            K.fill(self.amax[1], 0.0)
            K.absmax(dy, self._am(1, n - 1))
            if any(l.get("dy_s16") for l in self.layers):
                K.fill(self.pq, 0.0)
            if self.layers[-1].get("dy_s16") and self.layers[-1].get("ws_mode") == getattr(K, "conv_precision", 0):
                # no LayerNorm kernel writes the head's gradient: one conversion pass, and the last layer's dgrad and filter gradient
                # stage it by DMA like every other layer's
                if getattr(self, "_dctx16", None) is None:
                    self._dctx16 = torch.empty_like(dy)
                K.presplit16(dy, self._dctx16, self._am(1, n - 1))
                dy, dy_s16 = self._dctx16, True
        cur = -1                                   # dY buffer holding dy (-1: the caller's dctx)
        c3_fused = None                            # dA of layer 0 when its LayerNorm backward ran without the apply pass (below)
        for j in range(n - 1, -1, -1):
            lay = self.layers[j]
            prv = self.layers[j - 1] if j else None

            def wgrad(dy=dy, j=j, lay=lay, prv=prv, dy_s16=dy_s16):
                if j == 0 and c3_fused is not None:
                    # conv1_1: dy was never written - the filter gradient computes it from the LayerNorm backward's operands
                    K.conv_c3_wgrad_ln(self.images, lay["y"], c3_fused, lay["gamma"], lay["beta"], lay["stats"], self._ln0_means, lay["gw"])
                    return
                if prv is not None and prv.get("fused_now"):
                    # the input activation was never written: the wgrad kernel applies LayerNorm + ELU to the producing layer's y
                    K.conv_wgrad(prv["y"], dy, lay["gw"], lay["s"], self._am(0, j - 1), self._am(1, j), ln=(prv["stats"], prv["gamma"], prv["beta"]),
                                 **({"dy_s16": True} if dy_s16 else {}))
                else:
                    x_in = self.images if j == 0 else prv["a"]
                    x_s16 = bool(j and prv.get("a_s16_now"))
                    if x_s16 or dy_s16:
                        K.conv_wgrad(x_in, dy, lay["gw"], lay["s"], self._am(0, j - 1) if j else None, self._am(1, j), x_s16=x_s16, dy_s16=dy_s16)
                    elif f16:
                        K.conv_wgrad(x_in, dy, lay["gw"], lay["s"], self._am(0, j - 1) if j else None, self._am(1, j))
                    else:
                        K.conv_wgrad(x_in, dy, lay["gw"], lay["s"])
            def wgrad_on_side(cur=cur, lay=lay):
                with torch.cuda.stream(side):
                    wgrad()
                    if not lay["has_ln"]:
                        # last conv: BiasAddGrad = column sums of dy - read only by the optimiser, so it leaves the chain with the
                        # filter gradient (the head's f32 gradient: a tensor of the caller that nothing overwrites before the join)
                        K.colsum(dy_f32.view(-1, lay["cout"]), lay["gb"], False)
                    if cur >= 0:
                        reader_done[cur] = torch.cuda.Event()
                        reader_done[cur].record(side)
            late = side is not None and j > 0 and bool(getattr(K, "wgrad_late", False))
            if side is None:
                wgrad()
            elif not late:
                side.wait_stream(main)            # dy_j (and its amax word) are complete
                wgrad_on_side()
            if not lay["has_ln"] and side is None:
                # last conv: BiasAddGrad = column sums of dy (LN layers get theirs from ln_elu_bwd below)
                K.colsum(dy_f32.view(-1, lay["cout"]), lay["gb"], False)
            if j == 0:
                break
            prev = self.layers[j - 1]
            numel = prev["a"].numel()
            dA = self._dA[:numel].view(prev["out_shape"])
            nxt = (cur + 1) % len(dybufs)
            if reader_done[nxt] is not None:      # the buffer LayerNorm backward is about to overwrite may still be read by a wgrad
                main.wait_event(reader_done[nxt])
                reader_done[nxt] = None
            dYp = dybufs[nxt][:numel].view(prev["out_shape"])
            ws = lay["ws_bwd"] if (lay["ws_bwd"] is not None and lay.get("ws_mode") == getattr(K, "conv_precision", 0)) else None
            if dy_s16:
                assert ws is not None
                K.conv_dgrad(dy, lay["w"], dA, lay["s"], ws, self._am(1, j), self._am(2, j), lay["ws_layout_bwd"], dy_s16=True)
            elif ws is not None or f16:
                assert not (ws is not None and lay["ws_layout_bwd"] == 4 and lay["cin"] % 128), "four-block producer / consumer tiles need a pre-split dy"
                K.conv_dgrad(dy, lay["w"], dA, lay["s"], ws, self._am(1, j), self._am(2, j), lay["ws_layout_bwd"] if ws is not None else 0)
            else:
                K.conv_dgrad(dy, lay["w"], dA, lay["s"])
            if late:
                # option wgrad_late: the filter gradient starts when dgrad_j has FINISHED, i.e. beside the HBM-bound LayerNorm backward
                # of layer j - 1 instead of beside dgrad_j (both MFMA-bound)
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                wgrad_on_side()
            nxt_s16 = bool(prev.get("dy_s16")) and prev.get("ws_mode") == getattr(K, "conv_precision", 0) and self._ln_fin is not None
            if (j == 1 and prev["cin"] == 3 and prev["region"] is None and self._ln_fin is not None and hasattr(K, "conv_c3_wgrad_ln")
                    and getattr(K, "c3_ln_bwd_fused", False) and prev["out_shape"][3] == 32):
                # conv1_1's LayerNorm: the reductions only.  Its dy has ONE consumer, conv1_1's filter gradient, which computes it
                # from (y, dA) itself (sgg_conv2d_nhwc_wgrad_c3_ln): no apply pass, no dy tensor.  dA stays untouched until then
                # (the loop ends with that filter gradient; the next backward on this encoder is ordered behind it).
                if getattr(self, "_ln0_means", None) is None:
                    self._ln0_means = torch.empty((B, 2), device=dA.device, dtype=dA.dtype)
                K.ln_elu_bwd_sums(prev["y"], dA, prev["gamma"], prev["beta"], prev["stats"], self._ln0_means, prev["ln_ws"])
                c3_fused = dA
                dy, cur, dy_s16 = None, nxt, False
                continue
            if nxt_s16:
                K.ln_elu_bwd(prev["y"], dA, prev["gamma"], prev["beta"], prev["stats"], dYp, None, None, None, self._am(1, j - 1),
                             region=prev["region"], ws=prev["ln_ws"], out_s16=True, pq=self.pq[j - 1])
            elif self._ln_fin is not None:
                K.ln_elu_bwd(prev["y"], dA, prev["gamma"], prev["beta"], prev["stats"], dYp, None, None, None, self._am(1, j - 1) if f16 else None,
                             region=prev["region"], ws=prev["ln_ws"])
            elif prev["region"] is not None:
                K.ln_elu_bwd(prev["y"], dA, prev["gamma"], prev["beta"], prev["stats"], dYp, prev["ggamma"], prev["gbeta"], prev["gb"],
                             *([self._am(1, j - 1)] if f16 else []), region=prev["region"])
            elif f16:
                K.ln_elu_bwd(prev["y"], dA, prev["gamma"], prev["beta"], prev["stats"], dYp, prev["ggamma"], prev["gbeta"], prev["gb"],
                             self._am(1, j - 1))
            else:
                K.ln_elu_bwd(prev["y"], dA, prev["gamma"], prev["beta"], prev["stats"], dYp, prev["ggamma"], prev["gbeta"], prev["gb"])
            dy, cur, dy_s16 = dYp, nxt, nxt_s16
        if self._ln_fin is not None:
            K.ln_bwd_finalize(self._ln_fin)       # dgamma, dbeta and the conv bias gradients of all eleven LayerNorms
        if side is not None:
            main.wait_stream(side)                # every filter gradient is complete before the optimiser / all-reduce reads it
