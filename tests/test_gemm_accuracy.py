"""Direct accuracy bound of the split-precision linear layers (VERDICT r1 "weak" #9).

The scalar MLPs of an fp32 model run on the bf16 matrix cores: every fp32 operand is split by TRUNCATION into three
bf16 pieces and the six leading cross products are accumulated in fp32 (aa_gemm.hip, "bf16x3").  The bench line labels
that arithmetic "f32"; this test is what lets the label stand: on adversarial operands the bf16x3 kernels (single
layer and fused chain) may not be further from an fp64 product than the native fp32-input MFMA kernel is
(2x + one ulp of the row scale), and both must be fp32-class in absolute terms.

Operands: (i) N(0,1); (ii) magnitudes spanning 2^-20 .. 2^20 within every row; (iii) cancelling pairs
(a_k w_k + a_k' w_k' = tiny difference of large terms).  K covers the widths of the reverse pass (64 .. 448).
GPU: the real kernels.  CPU: the same kernels in the test-only emulation build (bit-level MFMA model)."""
import ctypes as C

import numpy as np
import pytest
import torch

from allegro_amd import _lib


def _operands(kind, M, K, N, rng):
    a = rng.standard_normal((M, K))
    w = rng.standard_normal((K, N)) / np.sqrt(K)
    if kind == "range":
        a *= np.exp2(rng.integers(-20, 21, size=(M, K)))
        w *= np.exp2(rng.integers(-6, 7, size=(K, N)))
    elif kind == "cancel":
        # pairs (k, k+1): a_{k+1} = -a_k (1 + d), w_{k+1} = w_k: the sum keeps only a_k w_k d, d ~ 1e-4
        a[:, 1::2] = -a[:, 0::2] * (1.0 + 1e-4 * rng.standard_normal((M, K // 2)))
        w[1::2, :] = w[0::2, :]
    return a.astype(np.float32), w.astype(np.float32)


def _run(lib, kernel, a, w, dev):
    M, K = a.shape
    N = w.shape[1]
    at = torch.tensor(a, device=dev)
    ct = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    wh = np.ascontiguousarray(w)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    lib.check(lib.lib.aa_debug_gemm_f32(kernel, M, K, N, at.data_ptr(), wh.ctypes.data_as(C.c_void_p), ct.data_ptr(), stream),
              "aa_debug_gemm_f32")
    return ct.cpu().numpy().astype(np.float64)


def _check(lib, dev, shapes, M, relative=True):
    rng = np.random.default_rng(17)
    worst = {}
    for kind in ("normal", "range", "cancel"):
        for K, N in shapes:
            a, w = _operands(kind, M, K, N, rng)
            ref = a.astype(np.float64) @ w.astype(np.float64)
            # row scale: sum_k |a_k w_k| -- what one rounding error of any term is relative to
            scale = np.abs(a.astype(np.float64)) @ np.abs(w.astype(np.float64))
            err = {}
            kernels = (("bf16x3", 0), ("fp32_mfma", 1)) + ((("chain", 2),) if K <= 384 else ())  # (a chain layer holds <= 12 k chunks)
            for name, kern in kernels:
                got = _run(lib, kern, a, w, dev)
                assert np.isfinite(got).all(), (name, kind, K, N)
                err[name] = np.max(np.abs(got - ref) / scale)
            ulp = 2.0 ** -23
            for name in [n for n, _ in kernels if n != "fp32_mfma"]:
                # (1) not worse than the hardware's own fp32 matrix path, (2) fp32-class in absolute terms: a K-term
                # fp32 accumulation is allowed ~sqrt(K) ulp of the row scale; the split adds < 3 * 2^-24 per product
                if relative:
                    assert err[name] <= 2.0 * err["fp32_mfma"] + ulp, (name, kind, K, N, err)
                assert err[name] <= (4.0 + 0.5 * np.sqrt(K)) * ulp, (name, kind, K, N, err)
            worst[(kind, K, N)] = err
    return worst


def test_bf16x3_gemm_error_bound_emulated():
    from tests.hip_utils import emu_lib

    # (the emulator rounds after every single product of an MFMA, i.e. 6x more roundings than the fp32 path it is compared
    #  with; the hardware sums the 16 products of an instruction before rounding -- so only the absolute fp32-class
    #  bound is asserted here and the comparison with the native fp32 MFMA kernel is made on the GPU)
    _check(emu_lib(), torch.device("cpu"), [(64, 64), (192, 64), (448, 64)], M=40, relative=False)


@pytest.mark.gpu
def test_bf16x3_gemm_error_bound_on_gpu():
    worst = _check(_lib.load(), torch.device("cuda:0"), [(64, 64), (64, 256), (128, 64), (192, 64), (256, 64), (448, 64)], M=1000)
    for k, v in worst.items():
        print(k, {n: f"{e:.2e}" for n, e in v.items()})
