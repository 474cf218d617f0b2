"""Pins of the third-party leaf conventions that the reference itself stores no golden vectors for (SURVEY §8c,
"parity unpinned" residual): Clebsch-Gordan values against sympy's exact Wigner symbols, and the real-basis 3j tensors
against the spherical harmonics they have to be consistent with -- D_l(R) is derived from the harmonics by least
squares (Y_l(R r) = D_l(R) Y_l(r)) and the 3j tensor must be invariant under D (x) D (x) D.  Also: the product's own
tables (allegro_amd/o3.py) and the oracle's shim (oracle/shim/e3nn) are two independent derivations and must agree."""
import itertools
import os
import sys

import numpy as np
import pytest
import torch

from allegro_amd import o3

TRIPLES = [(a, b, c) for a, b, c in itertools.product(range(4), repeat=3) if abs(a - b) <= c <= a + b]


def test_su2_clebsch_gordan_against_sympy_exact():
    from sympy import S
    from sympy.physics.wigner import clebsch_gordan

    for j1, j2, j3 in TRIPLES:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                m3 = m1 + m2
                if abs(m3) > j3:
                    continue
                want = float(clebsch_gordan(S(j1), S(j2), S(j3), S(m1), S(m2), S(m3)))
                assert abs(o3.su2_cg(j1, m1, j2, m2, j3, m3) - want) < 1e-13, (j1, m1, j2, m2, j3, m3)


def _sh(vec, l_max):
    from oracle import restatement as R

    return R.spherical_harmonics_lmax3(torch.as_tensor(vec, dtype=torch.float64), l_max).numpy()


def _rotation(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


def _wigner_d_from_sh(l, rot, rng):
    pts = rng.standard_normal((64, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    sl = slice(l * l, (l + 1) * (l + 1))
    y, yr = _sh(pts, 3)[:, sl], _sh(pts @ rot.T, 3)[:, sl]
    d, res, *_ = np.linalg.lstsq(y, yr, rcond=None)   # yr = y @ d  =>  Y(R r) = d^T Y(r)
    assert np.abs(y @ d - yr).max() < 1e-12, "the harmonics of degree l do not form a representation"
    return d.T


def test_spherical_harmonics_are_component_normalised_polynomials():
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((200, 3))
    y = _sh(pts, 3)                                   # normalize=True: direction only
    assert np.allclose(y, _sh(3.7 * pts, 3))
    for l in range(4):
        assert np.allclose((y[:, l * l:(l + 1) * (l + 1)] ** 2).sum(1), 2 * l + 1)   # "component" normalisation
    unit = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    assert np.allclose(y[:, 1:4], np.sqrt(3) * unit)  # l = 1 is (x, y, z) in this order (e3nn: D^1(R) = R)


@pytest.mark.parametrize("l1,l2,l3", TRIPLES)
def test_real_3j_tensors_are_invariant_under_the_representation_of_the_harmonics(l1, l2, l3):
    rng = np.random.default_rng(7 + 16 * l1 + 4 * l2 + l3)
    rot = _rotation(rng)
    d1, d2, d3 = (_wigner_d_from_sh(l, rot, rng) for l in (l1, l2, l3))
    for d in (d1, d2, d3):
        assert np.allclose(d @ d.T, np.eye(d.shape[0]), atol=1e-10)  # orthogonal: real irreps in an orthonormal basis
    w = o3.wigner_3j(l1, l2, l3)
    assert np.abs(np.einsum("ai,bj,ck,ijk->abc", d1, d2, d3, w) - w).max() < 1e-10
    # and the coupling of two harmonics of one direction is again the harmonic (where it does not vanish by parity)
    pts = rng.standard_normal((16, 3))
    y = _sh(pts, 3)
    ya, yb, yc = (y[:, l * l:(l + 1) * (l + 1)] for l in (l1, l2, l3))
    coupled = np.einsum("ni,nj,ijk->nk", ya, yb, w)
    if (l1 + l2 + l3) % 2 == 0:
        ratio = (coupled * yc).sum(1) / (yc * yc).sum(1)
        assert np.abs(coupled - ratio[:, None] * yc).max() < 1e-10 and np.ptp(ratio) < 1e-10 and abs(ratio[0]) > 1e-3
    else:
        assert np.abs(coupled).max() < 1e-10


def test_product_tables_agree_with_the_oracle_shim():
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "shim")
    sys.path.insert(0, shim)
    try:
        from e3nn.o3._wigner import wigner_3j as shim_w3j
    finally:
        sys.path.remove(shim)
    for l1, l2, l3 in TRIPLES:
        a, b = o3.wigner_3j(l1, l2, l3), shim_w3j(l1, l2, l3, dtype=torch.float64).numpy()
        assert np.abs(a - b).max() < 1e-12, (l1, l2, l3)
