"""ORACLE SHIM: real spherical harmonics in the e3nn convention.

Reference call site: allegro/nn/tensorembed.py:55-57,92
(`SphericalHarmonics(irreps, normalize=True, normalization="component")`).

Convention restated from e3nn (absent here): polynomials of (x, y, z) with y as
the polar axis, m ordered -l..l, so that Y_1 = (x, y, z); "component"
normalisation means sum_m Y_lm^2 = 2l+1 on unit vectors.  Equivalent to the
standard (z-polar, no Condon-Shortley sign) real harmonics evaluated at
(x_std, y_std, z_std) = (z, x, y).  Implemented generically by the associated
Legendre recursion (an independent code path from the explicit l<=3 polynomials
used in oracle/restatement.py and in the HIP kernels).  PARITY UNPINNED vs e3nn.
"""
import math

import torch

from ._irreps import Irreps


def _sh_component(lmax: int, x: torch.Tensor, y: torch.Tensor, z: torch.Tensor):
    """All l<=lmax, component normalised, unit-vector input. Returns list over l of [...,2l+1]."""
    # std coords: polar axis = y, azimuth measured from z towards x
    xs, ys, zs = z, x, y
    ct = zs  # cos(theta)
    # (sin(theta))^m * cos(m phi), (sin theta)^m * sin(m phi) via complex powers of (xs + i ys)
    cm = [torch.ones_like(x)]
    sm = [torch.zeros_like(x)]
    for m in range(1, lmax + 1):
        cm.append(cm[-1] * xs - sm[-1] * ys)
        sm.append(sm[-1] * xs + cm[-2] * ys)
    # P~_l^m(ct) / sin^m(theta), no Condon-Shortley phase
    P = {}
    for m in range(0, lmax + 1):
        pmm = float(math.prod(range(2 * m - 1, 0, -2))) if m > 0 else 1.0
        P[(m, m)] = torch.full_like(x, pmm)
        if m + 1 <= lmax:
            P[(m + 1, m)] = ct * (2 * m + 1) * P[(m, m)]
        for l in range(m + 2, lmax + 1):
            P[(l, m)] = ((2 * l - 1) * ct * P[(l - 1, m)] - (l + m - 1) * P[(l - 2, m)]) / (l - m)
    out = []
    for l in range(lmax + 1):
        comps = []
        for m in range(-l, l + 1):
            am = abs(m)
            n = math.sqrt((2 * l + 1) * math.factorial(l - am) / math.factorial(l + am))
            if m == 0:
                comps.append(n * P[(l, 0)])
            elif m > 0:
                comps.append(math.sqrt(2) * n * P[(l, am)] * cm[am])
            else:
                comps.append(math.sqrt(2) * n * P[(l, am)] * sm[am])
        out.append(torch.stack(comps, dim=-1))
    return out


def spherical_harmonics(ls, vec: torch.Tensor, normalize: bool, normalization: str = "component"):
    if isinstance(ls, int):
        ls = [ls]
    if normalize:
        vec = torch.nn.functional.normalize(vec, dim=-1)
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    if not normalize:
        raise NotImplementedError("oracle shim only implements normalize=True")
    per_l = _sh_component(max(ls), x, y, z)
    outs = []
    for l in ls:
        yl = per_l[l]
        if normalization == "component":
            pass
        elif normalization == "norm":
            yl = yl / math.sqrt(2 * l + 1)
        elif normalization == "integral":
            yl = yl / math.sqrt(4 * math.pi)
        else:
            raise ValueError(normalization)
        outs.append(yl)
    return torch.cat(outs, dim=-1)


class SphericalHarmonics(torch.nn.Module):
    def __init__(self, irreps_out, normalize: bool, normalization: str = "integral", irreps_in=None):
        super().__init__()
        if isinstance(irreps_out, int):
            irreps_out = Irreps.spherical_harmonics(irreps_out)
        self.irreps_out = Irreps(irreps_out)
        for mul, ir in self.irreps_out:
            assert ir.p == (-1) ** ir.l, "SH irreps must have parity (-1)^l"
        self._ls = [ir.l for mul, ir in self.irreps_out for _ in range(mul)]
        self.normalize = normalize
        self.normalization = normalization

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return spherical_harmonics(self._ls, x, self.normalize, self.normalization)
