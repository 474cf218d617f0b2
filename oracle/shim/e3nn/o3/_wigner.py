"""ORACLE SHIM: real-basis Wigner 3j symbols.

Reference call site: allegro/nn/_strided/_contract.py:95 (`wigner_3j(l1, l2, l3)`),
then scaled by sqrt(2*l_out+1) at :110,115.

Algorithm restated from e3nn (version transitively pinned by nequip>=0.13.0,
pyproject.toml:15-17; source NOT under /root/reference):
    C_real[j,l,m] = sum_{i,k,n} Q1[i,j] Q2[k,l] conj(Q3)[n,m] <l1 i l2 k | l3 n>,
    normalised to Frobenius norm 1, where Q_l is the real->complex change of
    basis including the (-i)^l phase that makes the result real.
su(2) Clebsch-Gordan coefficients come from sympy (exact rationals/surds), which
is an independent source from the product's float Racah implementation
(allegro_amd/o3.py).  PARITY UNPINNED vs e3nn (absent); pinned by identities in
tests/test_conventions.py (norm 1, w3j(l,l,0)=delta/sqrt(2l+1), nnz counts of
SURVEY.md §8c, joint SH/CG equivariance).
"""
import functools
import math

import numpy as np
import torch


def change_basis_real_to_complex(l: int) -> np.ndarray:
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@functools.lru_cache(maxsize=None)
def _su2_cg(l1: int, l2: int, l3: int) -> np.ndarray:
    from sympy import S
    from sympy.physics.quantum.cg import CG

    out = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                out[l1 + m1, l2 + m2, l3 + m3] = float(
                    CG(S(l1), S(m1), S(l2), S(m2), S(l3), S(m3)).doit()
                )
    return out


@functools.lru_cache(maxsize=None)
def _so3_cg(l1: int, l2: int, l3: int) -> np.ndarray:
    q1 = change_basis_real_to_complex(l1)
    q2 = change_basis_real_to_complex(l2)
    q3 = change_basis_real_to_complex(l3)
    c = _su2_cg(l1, l2, l3).astype(np.complex128)
    c = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), c)
    assert np.abs(c.imag).max() < 1e-10
    c = c.real
    c = c / np.linalg.norm(c)
    c[np.abs(c) < 1e-14] = 0.0
    return c


def wigner_3j(l1: int, l2: int, l3: int, dtype=None, device=None) -> torch.Tensor:
    assert abs(l2 - l3) <= l1 <= l2 + l3
    if dtype is None:
        dtype = torch.get_default_dtype()
    return torch.tensor(_so3_cg(l1, l2, l3), dtype=dtype, device=device)
