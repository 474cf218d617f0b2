"""ORACLE SHIM: radial embedding leaves (allegro_models.py:153-157,275-278; scalarembed.py:60-66).

Restated from memory of nequip -- PARITY UNPINNED (exact Bessel prefactor unverifiable here):
  NORM_LENGTH x = r / r_max(type_i,type_j);  cutoff f_p(x) (polynomial envelope, 0 for x>=1);
  b_n(x) = sin(n*pi*x)/x, n=1..num_bessels;  embedding = b_n(x) * f_p(x).
"""
import math

import torch

from e3nn.o3._irreps import Irreps

from ..data import AtomicDataDict
from . import GraphModuleMixin


class PolynomialCutoff(torch.nn.Module):
    def __init__(self, p: float = 6):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        p = self.p
        out = 1.0 - ((p + 1.0) * (p + 2.0) / 2.0) * torch.pow(x, p) + p * (p + 2.0) * torch.pow(x, p + 1.0) \
            - (p * (p + 1.0) / 2.0) * torch.pow(x, p + 2.0)
        return out * (x < 1.0)


class EdgeLengthNormalizer(GraphModuleMixin, torch.nn.Module):
    def __init__(self, r_max, type_names, per_edge_type_cutoff=None, irreps_in=None):
        super().__init__()
        self.r_max = float(r_max)
        self.num_types = len(type_names)
        self._per_edge_type = per_edge_type_cutoff is not None
        rmax = torch.full((self.num_types, self.num_types), self.r_max, dtype=torch.get_default_dtype())
        if self._per_edge_type:
            for ci, cname in enumerate(type_names):
                if cname not in per_edge_type_cutoff:
                    continue
                v = per_edge_type_cutoff[cname]
                for ni, nname in enumerate(type_names):
                    if isinstance(v, dict):
                        if nname in v:
                            rmax[ci, ni] = float(v[nname])
                    else:
                        rmax[ci, ni] = float(v)
            assert float(rmax.max()) <= self.r_max + 1e-12
        self.register_buffer("rmax_recip", 1.0 / rmax)
        self._init_irreps(irreps_in=irreps_in, irreps_out={AtomicDataDict.NORM_LENGTH_KEY: Irreps("1x0e")})

    def forward(self, data):
        data = AtomicDataDict.with_edge_vectors_(data, with_lengths=True)
        r = data[AtomicDataDict.EDGE_LENGTH_KEY]
        if self._per_edge_type:
            et = torch.index_select(data[AtomicDataDict.ATOM_TYPE_KEY].reshape(-1), 0,
                                    data[AtomicDataDict.EDGE_INDEX_KEY].reshape(-1)).view(2, -1)
            data[AtomicDataDict.EDGE_TYPE_KEY] = et
            recip = self.rmax_recip[et[0], et[1]]
        else:
            recip = self.rmax_recip[0, 0]
        data[AtomicDataDict.NORM_LENGTH_KEY] = (r * recip).unsqueeze(-1)
        return data


class BesselEdgeLengthEncoding(GraphModuleMixin, torch.nn.Module):
    def __init__(self, cutoff, num_bessels: int = 8, trainable: bool = False,
                 edge_invariant_field=AtomicDataDict.EDGE_EMBEDDING_KEY, irreps_in=None):
        super().__init__()
        self.cutoff = cutoff
        self.num_bessels = num_bessels
        self.out_field = edge_invariant_field
        w = torch.linspace(1.0, num_bessels, num_bessels).unsqueeze(0) * math.pi
        if trainable:
            self.bessel_weights = torch.nn.Parameter(w)
        else:
            self.register_buffer("bessel_weights", w)
        self._init_irreps(irreps_in=irreps_in, irreps_out={
            edge_invariant_field: Irreps([(num_bessels, (0, 1))]),
            AtomicDataDict.EDGE_CUTOFF_KEY: Irreps("1x0e")})

    def forward(self, data):
        x = data[AtomicDataDict.NORM_LENGTH_KEY]
        bessel = torch.sin(self.bessel_weights * x) / x
        cut = self.cutoff(x)
        data[AtomicDataDict.EDGE_CUTOFF_KEY] = cut
        data[self.out_field] = bessel * cut
        return data


class AddRadialCutoffToData(GraphModuleMixin, torch.nn.Module):
    def __init__(self, cutoff, irreps_in=None):
        super().__init__()
        self.cutoff = cutoff
        self._init_irreps(irreps_in=irreps_in, irreps_out={AtomicDataDict.EDGE_CUTOFF_KEY: Irreps("1x0e")})

    def forward(self, data):
        if AtomicDataDict.EDGE_CUTOFF_KEY not in data:
            data[AtomicDataDict.EDGE_CUTOFF_KEY] = self.cutoff(data[AtomicDataDict.NORM_LENGTH_KEY])
        return data
