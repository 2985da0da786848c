"""Generates tests/golden/config1_golden.npz from the CPU oracle (oracle/sgg_oracle.py) on BASELINE.json configs[0]
(8 x 64x64 N(0,1) images seed 0, vocab 50, labels seed 1, noise seed 2+k, alpha seed 1000+k, weights seed 3).

The reference cannot be run or imported in this pipeline, so these vectors are outputs of the
restatement, not of the reference: they pin the oracle against accidental change and give the GPU path a
checked-in target.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sgg_oracle as O  # noqa: E402


def compute(B=8, S=64, V=50, small=False):
    """small: keep the file small at full layer shapes (no feature map, only short gradient tensors)."""
    torch.set_num_threads(4)
    gp, dp = O.init_params("G", V, S), O.init_params("D", V, S)
    images, labels, onehot = O.synth_batch(B, S, V)
    noise0, noise1, alpha = O.synth_noise(B, 0), O.synth_noise(B, 1), O.synth_alpha(B, 0)
    out = {"labels": labels.numpy()}
    fake0 = O.generator_forward(gp, images, noise0)
    out["g_logits_step0"] = fake0.numpy()
    out["g_tokens_step0"] = O.argmax_tokens(fake0).numpy()
    out["d_real_step0"] = O.discriminator_forward(dp, onehot, images).numpy()
    out["d_fake_step0"] = O.discriminator_forward(dp, fake0, images).numpy()
    feat = O.encoder(gp, images)
    out["g_downsampled_sample0"] = feat[0].numpy() if not small else feat[0, ::4, ::4, ::16].numpy()
    d_adam, g_adam = O.new_adam_state(dp), O.new_adam_state(gp)
    cost, aux, dgrads = O.d_step(gp, dp, d_adam, 1, images, onehot, noise0, alpha)
    out["disc_cost"] = np.float32(cost)
    out["slopes"] = aux["slopes"].detach().numpy()
    out["gp"] = np.float32(aux["gp"].detach())
    for n in ("conv2d/kernel", "conv2d_13/bias", "LayerNorm_5/gamma", "layer_norm_basic_lstm_cell/state/beta", "decoder/kernel"):
        out["dgrad/" + n] = dgrads[n].numpy()
    out["dgrad_l1/W"] = np.float32(dgrads["W"].abs().sum())
    out["dgrad_l1/attention_perceptron/kernel"] = np.float32(dgrads["attention_perceptron/kernel"].abs().sum())
    gcost, gaux, ggrads = O.g_step(gp, dp, g_adam, 1, images, noise1)
    out["gen_cost"] = np.float32(gcost)
    out["g_tokens_step1"] = O.argmax_tokens(gaux["fake"]).numpy()
    out["g_margin_step1"] = np.float32(O.top2_margin(gaux["fake"]))
    for n in ("conv2d/kernel", "LayerNorm_12/beta", "decoder/bias", "layer_norm_basic_lstm_cell/input/gamma"):
        out["ggrad/" + n] = ggrads[n].numpy()
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    # BASELINE.json configs[0] (the file name predates the 0-based numbering)
    todo = [("config1_golden.npz", dict(B=8, S=64, V=50)),
            # two samples at BASELINE.json configs[1] layer shapes (224x224, vocab 1000)
            ("configs1_shape_b2_golden.npz", dict(B=2, S=224, V=1000, small=True)),
            # two samples at configs[4] layer shapes (448x448: L = 784, attention W [401920+512, 784])
            ("configs4_shape_b2_golden.npz", dict(B=2, S=448, V=50, small=True)),
            # configs[3] vocabulary (70 000: decoder [512,70000], embedding [70000,300]) on small images
            ("configs3_vocab_b2_golden.npz", dict(B=2, S=64, V=70000, small=True)),
            # the reference's real-data resolution (train.py:171 resizes to 221x221: odd maps 221, 111, then 56, 28, 14)
            ("realdata_221px_b2_golden.npz", dict(B=2, S=221, V=1000, small=True))]
    only = sys.argv[1:]
    for name, kw in todo:
        if only and name not in only:
            continue
        o = compute(**kw)
        if kw.get("small"):      # [B,3,V] logits at V = 70 000 are kept as a strided sample + the arg-maxed tokens
            if o["g_logits_step0"].shape[-1] > 2000:
                o["g_logits_step0"] = o["g_logits_step0"][..., ::97]
            o.pop("dgrad/decoder/kernel", None)
        path = os.path.join(here, name)
        np.savez_compressed(path, **o)
        print(path, os.path.getsize(path), "bytes")
