"""TEST INFRASTRUCTURE: a module-structured stand-in for the reference's `AllegroModel` that can travel to the GPU box.

The reference model cannot be imported there (no /root/reference, no nequip/e3nn), but the operator-seam tests need a
`torch.nn.Module` whose tensor-product layers are *Contracter modules* so that the `enable_HipContracter` modifier has
something to swap.  `EagerContracter` carries the reference Contracter's attributes, parameter / buffer names and
forward signature (allegro/nn/_strided/_contract.py:33-211) around the oracle's eager contraction;
`ModuleAllegro` evaluates oracle/restatement.py with those modules in the layer slots and keeps the reference's
state_dict keys (`func.allegro.tps.{l}.weights|w3j`).  Pinned to the reference through the golden fixtures: before the
swap it must reproduce them too (checked by the tests that use it)."""
import math

import torch

from allegro_amd import o3
from allegro_amd.nn import allegro_layer_irreps
from oracle import restatement as R


class EagerContracter(torch.nn.Module):
    def __init__(self, irreps_in1, irreps_in2, irreps_out, mul, w3j, weights, path_channel_coupling, scatter_factor):
        super().__init__()
        self.irreps_in1, self.irreps_in2, self.irreps_out = irreps_in1, irreps_in2, irreps_out
        self.mul, self.instructions, self.path_channel_coupling = mul, None, path_channel_coupling
        self.scatter_factor, self.irrep_normalization = scatter_factor, "component"
        self.register_buffer("w3j", w3j.clone())
        self.weights = torch.nn.Parameter(weights.clone())

    def forward(self, x1, x2, idxs, scatter_dim_size):
        return R.contracter_forward(x1, x2, idxs, int(scatter_dim_size), self.weights, self.w3j,
                                    self.path_channel_coupling, self.scatter_factor)


class _Node(torch.nn.Module):
    pass


class ModuleAllegro(torch.nn.Module):
    """forward(data) -> {"atomic_energy", "total_energy", "forces"} with autograd forces (ForceStressOutput)."""

    def __init__(self, cfg: dict, sd: dict, dtype):
        super().__init__()
        self.cfg = dict(cfg)
        self.sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()
                   if not k.startswith("allegro.tps.")}
        L, u, l_max = cfg["num_layers"], cfg["num_tensor_features"], cfg["l_max"]
        irreps = allegro_layer_irreps(l_max, cfg.get("parity", True), L)
        env = o3.Irreps.spherical_harmonics(l_max, p=-1)
        self.func = _Node()
        self.func.allegro = _Node()
        self.func.allegro.tps = torch.nn.ModuleList([
            EagerContracter(str(irreps[l]), str(env), str(irreps[l + 1]), u, sd[f"allegro.tps.{l}.w3j"].to(dtype),
                            sd[f"allegro.tps.{l}.weights"].to(dtype), cfg.get("tp_path_channel_coupling", True),
                            1.0 / math.sqrt(float(cfg["avg_num_neighbors"]))) for l in range(L)])

    def to(self, *a, **k):
        super().to(*a, **k)
        self.sd = {kk: v.to(*a, **k) for kk, v in self.sd.items()}
        return self

    def forward(self, data, shift_vec=None):
        pos = data["pos"].detach().clone().requires_grad_(True)
        sd = dict(self.sd)
        for l, c in enumerate(self.func.allegro.tps):  # (only read for shapes / pruned irreps by the restatement)
            sd[f"allegro.tps.{l}.w3j"], sd[f"allegro.tps.{l}.weights"] = c.w3j, c.weights
        e_atom = R.allegro_energy(self.cfg, sd, pos, data["edge_index"], data["atom_types"], shift_vec,
                                  contracters=list(self.func.allegro.tps))
        (g,) = torch.autograd.grad(e_atom.sum(), pos, create_graph=self.training)  # training: forces stay differentiable
        return {"atomic_energy": e_atom, "total_energy": e_atom.sum().reshape(1, 1), "forces": -g}
