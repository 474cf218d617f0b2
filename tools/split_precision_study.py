"""Numerical study (CPU, numpy) for the GEMM operand formats of the matrix-core path: error of
  (a) bf16x3 (3 bf16 levels per operand, 6 cross products)   -- what the chain kernels do today
  (b) fp16x2 (2 fp16 levels per operand, 3 cross products), activations scaled per row by a power of two
against an fp64 reference, relative to the error of a plain fp32 dot product.  K = 64..256, activations ~ N(0,1)
with a wide per-row dynamic range (gradients), weights ~ U(-sqrt3, sqrt3)/sqrt(K).
    python tools/split_precision_study.py
"""
import numpy as np

rng = np.random.default_rng(0)


def trunc_bf16(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16x3(x):
    x = x.astype(np.float32)
    a = trunc_bf16(x)
    r = (x - a).astype(np.float32)
    b = trunc_bf16(r)
    c = trunc_bf16((r - b).astype(np.float32))
    return a, b, c


def split_fp16x2(x):
    x = x.astype(np.float32)
    h = x.astype(np.float16).astype(np.float32)
    l = (x - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, l


def mm32(a, b):  # products exact in fp32 (<= 22 significant bits), fp32 accumulation like the MFMA
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)


for K in (64, 128, 256):
    M, N = 4096, 64
    scale = 10.0 ** rng.uniform(-7, 1, size=(M, 1))  # rows spanning 8 decades (reverse-pass gradients)
    X = (rng.standard_normal((M, K)) * scale).astype(np.float32)
    W = (rng.uniform(-np.sqrt(3), np.sqrt(3), size=(K, N)) / np.sqrt(K)).astype(np.float32)
    ref = X.astype(np.float64) @ W.astype(np.float64)
    rowscale = np.abs(ref).max(axis=1, keepdims=True) + 1e-300
    err = lambda y: float(np.max(np.abs(y - ref) / rowscale))  # noqa: E731
    e32 = err(mm32(X, W))
    x1, x2, x3 = split_bf16x3(X)
    w1, w2, w3 = split_bf16x3(W)
    y = mm32(x1, w1) + mm32(x1, w2) + mm32(x2, w1) + mm32(x1, w3) + mm32(x2, w2) + mm32(x3, w1)
    e_b = err(y)
    # fp16x2: per-row power-of-two scale so that max|x| sits at 2^0..2^1 (exact to undo)
    ex = np.floor(np.log2(np.abs(X).max(axis=1, keepdims=True) + 1e-300))
    s = np.exp2(-ex).astype(np.float32)
    xh, xl = split_fp16x2(X * s)
    wh, wl = split_fp16x2(W)
    y2 = (mm32(xh, wh) + mm32(xh, wl) + mm32(xl, wh)) / s
    e_h = err(y2)
    xh0, xl0 = split_fp16x2(X)  # without the row scaling: shows why it is needed
    y3 = mm32(xh0, wh) + mm32(xh0, wl) + mm32(xl0, wh)
    e_h0 = err(y3)
    print(f"K={K:3d}  max row-relative error:  fp32 dot {e32:.2e} | bf16x3 (6 MFMA) {e_b:.2e} | "
          f"fp16x2 + row scale (3 MFMA) {e_h:.2e} | fp16x2 unscaled {e_h0:.2e}")
