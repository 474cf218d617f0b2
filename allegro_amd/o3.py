"""Minimal O(3) bookkeeping for the host side: irreps and real-basis Wigner 3j symbols.

The reference takes these from e3nn (`Irreps`, `wigner_3j`; call sites
allegro/nn/_strided/_contract.py:4-5,56-119 and allegro/nn/_allegro.py:43-160).  e3nn is not a
dependency here; this module implements the same conventions from the published formulas:
su(2) Clebsch-Gordan coefficients by Racah's closed form (exact integer arithmetic), rotated to the
real basis (m = -l..l, y polar, the (-i)^l phase making the tensor real), Frobenius-normalised.
Only init-time code uses it -- the numbers end up in the `w3j` buffers / device CG tables.
"""
import functools
import math
from fractions import Fraction
from typing import List, Tuple

import numpy as np


class Irrep:
    __slots__ = ("l", "p")

    def __init__(self, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                l, p = l.l, l.p
            elif isinstance(l, str):
                s = l.strip()
                l, p = int(s[:-1]), {"e": 1, "o": -1}[s[-1]]
            else:
                l, p = l
        self.l, self.p = int(l), int(p)
        assert self.l >= 0 and self.p in (1, -1)

    @property
    def dim(self):
        return 2 * self.l + 1

    def __mul__(self, other):
        other = Irrep(other)
        return [Irrep(l, self.p * other.p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __eq__(self, other):
        other = Irrep(other)
        return (self.l, self.p) == (other.l, other.p)

    def __hash__(self):
        return hash((self.l, self.p))

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"


class Irreps:
    """Ordered list of (mul, Irrep)."""

    def __init__(self, spec=None):
        items: List[Tuple[int, Irrep]] = []
        if isinstance(spec, Irreps):
            items = list(spec.items)
        elif isinstance(spec, str):
            for tok in spec.split("+"):
                tok = tok.strip()
                if not tok:
                    continue
                if "x" in tok:
                    m, ir = tok.split("x")
                    items.append((int(m), Irrep(ir)))
                else:
                    items.append((1, Irrep(tok)))
        elif spec is not None:
            for it in spec:
                if isinstance(it, (Irrep, str)):
                    items.append((1, Irrep(it)))
                else:
                    items.append((int(it[0]), Irrep(it[1])))
        self.items = items

    @staticmethod
    def spherical_harmonics(l_max: int, p: int = -1) -> "Irreps":
        return Irreps([(1, (l, p**l)) for l in range(l_max + 1)])

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return Irreps(self.items[i]) if isinstance(i, slice) else self.items[i]

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(x == ir for _, x in self.items)

    def __eq__(self, other):
        return self.items == Irreps(other).items

    @property
    def dim(self):
        return sum(m * ir.dim for m, ir in self.items)

    @property
    def num_irreps(self):
        return sum(m for m, _ in self.items)

    @property
    def lmax(self):
        return max(ir.l for _, ir in self.items)

    def offsets(self):
        out, o = [], 0
        for m, ir in self.items:
            out.append(o)
            o += m * ir.dim
        return out

    def __repr__(self):
        return "+".join(f"{m}x{ir}" for m, ir in self.items)


def _fact(n: int) -> int:
    return math.factorial(n)


def su2_cg(j1: int, m1: int, j2: int, m2: int, j3: int, m3: int) -> float:
    """<j1 m1 j2 m2 | j3 m3> by Racah's formula (integer j only)."""
    if m1 + m2 != m3 or not (abs(j1 - j2) <= j3 <= j1 + j2):
        return 0.0
    pref = Fraction((2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3), _fact(j1 + j2 + j3 + 1))
    pref *= _fact(j3 + m3) * _fact(j3 - m3) * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2)
    s = Fraction(0)
    for k in range(0, j1 + j2 - j3 + 1):
        args = [k, j1 + j2 - j3 - k, j1 - m1 - k, j2 + m2 - k, j3 - j2 + m1 + k, j3 - j1 - m2 + k]
        if min(args) < 0:
            continue
        d = 1
        for a in args:
            d *= _fact(a)
        s += Fraction((-1) ** k, d)
    return math.sqrt(float(pref)) * float(s)


def _real_to_complex(l: int) -> np.ndarray:
    """Rows: complex m=-l..l; columns: real components (sin-type for m<0, cos-type for m>0)."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    s = 1 / math.sqrt(2)
    for m in range(1, l + 1):
        q[l - m, l + m] = s
        q[l - m, l - m] = -1j * s
        q[l + m, l + m] = (-1) ** m * s
        q[l + m, l - m] = 1j * (-1) ** m * s
    q[l, l] = 1
    return (-1j) ** l * q


@functools.lru_cache(maxsize=None)
def _wigner_3j_np(l1: int, l2: int, l3: int) -> np.ndarray:
    c = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1), dtype=np.complex128)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                c[l1 + m1, l2 + m2, l3 + m3] = su2_cg(l1, m1, l2, m2, l3, m3)
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    r = np.einsum("ia,kb,nc,ikn->abc", q1, q2, np.conj(q3), c)
    assert np.abs(r.imag).max() < 1e-10, "real-basis 3j has an imaginary part"
    r = r.real
    r = r / np.linalg.norm(r)
    r[np.abs(r) < 1e-14] = 0.0
    return r


def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real-basis 3j tensor [2l1+1, 2l2+1, 2l3+1], Frobenius norm 1 (float64)."""
    assert abs(l1 - l2) <= l3 <= l1 + l2
    return _wigner_3j_np(l1, l2, l3).copy()


def tp_path_exists(irreps1: Irreps, irreps2: Irreps, ir_out) -> bool:
    ir_out = Irrep(ir_out)
    return any(ir_out in [x for x in ir1 * ir2] for _, ir1 in irreps1 for _, ir2 in irreps2)
