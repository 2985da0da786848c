"""Numerical error of Winograd F(2x2, 3x3) against the direct 3x3 convolution in the arithmetic of the default conv mode (operands
scaled by a power of two and split into two fp16 pieces with round-to-nearest, three products per MAC, f32 accumulation), on a
configs[1]-like layer (112 x 112 grid cut to 32 x 32, 128 -> 128 channels, He-normal weights, unit-variance activations).
CPU / NumPy only; feeds the plan in DESIGN.md section 9.   python scripts/study/winograd_error.py"""
import numpy as np

rng = np.random.default_rng(0)
H = W = 32
C, N = 128, 128
x = rng.standard_normal((H + 2, W + 2, C)).astype(np.float32)
x[0, :, :] = x[-1, :, :] = 0
x[:, 0, :] = x[:, -1, :] = 0
w = (rng.standard_normal((3, 3, C, N)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)


def split16(a):
    """a (f32) -> (hi, lo) fp16 pieces of a * 2^e, e such that max|a| * 2^e <= 2^14; returns pieces as f64 and the scale"""
    amax = float(np.abs(a).max())
    e = 14 - int(np.floor(np.log2(amax))) - 1 if amax > 0 else 0
    s = np.float32(2.0 ** e)
    t = (a * s).astype(np.float32)
    hi = t.astype(np.float16)
    lo = ((t - hi.astype(np.float32)) ).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64), float(s)


def contract(a, b):
    """sum_k a[..., k] * b[k, ...] in the split arithmetic: hi*hi + hi*lo + lo*hi, products exact, accumulation rounded to f32 per
    32-deep slab (the MFMA accumulates a K = 32 step exactly enough; slabs are added in f32)"""
    ah, al, sa = split16(a)
    bh, bl, sb = split16(b)
    K = a.shape[-1]
    acc = np.zeros(a.shape[:-1] + b.shape[1:], dtype=np.float32)
    for k0 in range(0, K, 32):
        sl = slice(k0, k0 + 32)
        part = ah[..., sl] @ bh[sl] + ah[..., sl] @ bl[sl] + al[..., sl] @ bh[sl]
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc / np.float32(sa * sb)


# reference: fp64 direct
ref = np.zeros((H, W, N))
for kh in range(3):
    for kw in range(3):
        ref += x[kh:kh + H, kw:kw + W, :].astype(np.float64) @ w[kh, kw].astype(np.float64)

# direct convolution in the split arithmetic (9 taps accumulated in f32)
direct = np.zeros((H, W, N), dtype=np.float32)
xa = x
for kh in range(3):
    for kw in range(3):
        direct = (direct + contract(xa[kh:kh + H, kw:kw + W, :].reshape(-1, C), w[kh, kw]).reshape(H, W, N)).astype(np.float32)

# Winograd F(2x2, 3x3): transforms in f32, the 16 products in the split arithmetic
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float32)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float32)
U = np.einsum("ia,abcn,jb->ijcn", G, w, G).astype(np.float32)                  # [4,4,C,N]
th, tw = H // 2, W // 2
tiles = np.zeros((th, tw, 4, 4, C), dtype=np.float32)
for i in range(th):
    for j in range(tw):
        tiles[i, j] = x[2 * i:2 * i + 4, 2 * j:2 * j + 4, :]
V = np.einsum("ia,tuabc,jb->tuijc", Bt, tiles, Bt).astype(np.float32)         # [th,tw,4,4,C]
M = np.zeros((th, tw, 4, 4, N), dtype=np.float32)
for i in range(4):
    for j in range(4):
        M[:, :, i, j, :] = contract(V[:, :, i, j, :].reshape(-1, C), U[i, j]).reshape(th, tw, N)
Y = np.einsum("ia,tuabn,jb->tuijn", At, M, At).astype(np.float32)             # [th,tw,2,2,N]
wino = Y.transpose(0, 2, 1, 3, 4).reshape(H, W, N)

scale = np.abs(ref).max()
for name, y in (("direct, split f16x3", direct), ("Winograd F(2x2,3x3), split f16x3", wino)):
    err = np.abs(y - ref)
    print("%-36s max |err| / max|ref| = %.2e   rms err / rms ref = %.2e" % (name, err.max() / scale, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())))
f32 = np.zeros((H, W, N), dtype=np.float32)
for kh in range(3):
    for kw in range(3):
        f32 = (f32 + (x[kh:kh + H, kw:kw + W, :].reshape(-1, C) @ w[kh, kw]).reshape(H, W, N)).astype(np.float32)
err = np.abs(f32 - ref)
print("%-36s max |err| / max|ref| = %.2e   rms err / rms ref = %.2e" % ("direct, numpy f32 matmul", err.max() / scale, np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())))
print("MACs per output pixel and channel pair: direct 9, Winograd 16 / 4 = 4 (2.25x fewer); transformed-input elements per input pixel: 4")
