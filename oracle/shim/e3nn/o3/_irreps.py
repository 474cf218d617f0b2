"""ORACLE SHIM: minimal `Irrep` / `Irreps` bookkeeping types.

Covers exactly the surface the reference touches:
`.dim .l .p .lmax .num_irreps .slices()`, `ir1 * ir2`, `ir in irreps`,
iteration as `(mul, ir)`, slicing, `repr`, `Irreps.spherical_harmonics`.
Used at: allegro/nn/_strided/_contract.py:56-119, allegro/nn/_allegro.py:43-160,
allegro/nn/_strided/_channels.py:36-37, allegro/model/allegro_models.py:76-86.
"""
import collections
import re


class Irrep(tuple):
    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                m = re.fullmatch(r"\s*(\d+)([eoy])\s*", l)
                assert m is not None, f"bad irrep {l!r}"
                ll = int(m.group(1))
                p = {"e": 1, "o": -1, "y": (-1) ** ll}[m.group(2)]
                l = ll
            elif isinstance(l, tuple):
                l, p = l
        assert isinstance(l, int) and l >= 0 and p in (-1, 1)
        return super().__new__(cls, (l, p))

    @property
    def l(self):  # noqa: E743
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self[0] + 1

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    def __mul__(self, other):
        other = Irrep(other)
        p = self.p * other.p
        return [Irrep(l, p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def is_scalar(self):
        return self.l == 0 and self.p == 1


class _MulIr(tuple):
    def __new__(cls, mul, ir):
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self.mul * self.ir.dim

    def __repr__(self):
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out = []
        if irreps is None:
            irreps = []
        if isinstance(irreps, Irrep):
            irreps = [(1, irreps)]
        if isinstance(irreps, str):
            if irreps.strip() != "":
                for tok in irreps.split("+"):
                    tok = tok.strip()
                    if "x" in tok:
                        mul, ir = tok.split("x")
                        out.append(_MulIr(int(mul), Irrep(ir.strip())))
                    else:
                        out.append(_MulIr(1, Irrep(tok)))
        else:
            for item in irreps:
                if isinstance(item, _MulIr):
                    out.append(item)
                elif isinstance(item, Irrep):
                    out.append(_MulIr(1, item))
                elif isinstance(item, str):
                    out.append(_MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p**l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(mi.dim for mi in self)

    @property
    def num_irreps(self):
        return sum(mi.mul for mi in self)

    @property
    def lmax(self):
        return max(mi.ir.l for mi in self)

    def slices(self):
        s, i = [], 0
        for mi in self:
            s.append(slice(i, i + mi.dim))
            i += mi.dim
        return s

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Irreps(super().__getitem__(i))
        return super().__getitem__(i)

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(mi.ir == ir for mi in self)

    def __add__(self, other):
        return Irreps(tuple(self) + tuple(Irreps(other)))

    def __repr__(self):
        return "+".join(repr(mi) for mi in self)

    def count(self, ir):
        ir = Irrep(ir)
        return sum(mi.mul for mi in self if mi.ir == ir)

    def randn(self, *size, device=None, dtype=None):
        import torch

        size = [self.dim if s == -1 else s for s in size]
        return torch.randn(*size, device=device, dtype=dtype)


_ = collections  # keep import for parity with e3nn's namedtuple-style API
