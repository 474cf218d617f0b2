"""Randomised configurations through the emulated kernels vs the oracle (fp64, 1e-9 relative): species count, l_max,
layers, multiplicities that are not multiples of 64 (general kernels) and 64 (specialised ones), MLP depths 1-2,
Bessel/spline bases of 4-12 functions, coupled/uncoupled path weights, per-type scales/shifts, unsorted edge lists;
energies, forces and the strain derivative.  The seeds are a fixed sample of a larger sweep (60 cases) run while
developing; the neighbor-list counterpart of that sweep is tests/test_neighbor_list.py."""
import numpy as np
import pytest
import torch

from tests.hip_utils import emu_lib
from allegro_amd import graph as G
from allegro_amd.nn import HipAllegroModel


@pytest.mark.parametrize("seed", [0, 1, 4, 9, 13, 20, 32, 34])
def test_random_configuration_vs_oracle(seed):
    from oracle import restatement as R

    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(4, 22))
    rc = float(rng.uniform(2.5, 3.5))
    box = float(rng.uniform(2 * rc + 0.2, 9.0))
    pos = rng.uniform(0, box, (n, 3))
    cell = np.eye(3) * box
    ei, shift = G.neighbor_list_pbc(pos, cell, rc)
    assert ei.shape[1] > 0
    T, l_max, L = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    u, S = int(rng.choice([2, 4, 8, 32, 64])), int(rng.choice([16, 32, 64]))
    spline, B = bool(rng.integers(0, 2)), int(rng.choice([4, 8, 12]))
    rce = ({"_target_": "allegro.nn.TwoBodySplineScalarEmbed", "num_splines": B, "spline_span": int(rng.integers(1, B + 1))}
           if spline else {"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": B})
    cfg = dict(type_names=["A", "B", "C"][:T], r_max=rc, l_max=l_max, num_layers=L, num_scalar_features=S,
               num_tensor_features=u, radial_chemical_embed=rce, radial_chemical_embed_dim=int(rng.choice([16, 32, 64])),
               scalar_embed_mlp_hidden_layers_depth=int(rng.integers(1, 3)),
               scalar_embed_mlp_hidden_layers_width=int(rng.choice([16, 64])),
               allegro_mlp_hidden_layers_depth=int(rng.integers(1, 3)),
               allegro_mlp_hidden_layers_width=int(rng.choice([32, 64])),
               readout_mlp_hidden_layers_depth=int(rng.integers(1, 3)),
               readout_mlp_hidden_layers_width=int(rng.choice([8, 64])),
               tp_path_channel_coupling=bool(rng.integers(0, 2)), avg_num_neighbors=float(max(1, ei.shape[1] / n)),
               seed=int(seed),
               per_type_energy_scales=[float(x) for x in rng.uniform(0.5, 2, T)] if rng.integers(0, 2) else None,
               per_type_energy_shifts=[float(x) for x in rng.uniform(-1, 1, T)] if rng.integers(0, 2) else None,
               model_dtype="float64")
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    types = torch.tensor(rng.integers(0, T, size=n))
    sv = torch.tensor(shift @ cell)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(seed))
    g = m.prepare_graph(torch.tensor(ei)[:, perm], types, n, sv[perm])
    e, f = m.energy_forces(torch.tensor(pos), g)
    w = m.virial(g)
    sd = {k[len("func."):]: v.detach() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos), torch.tensor(ei), types, sv)
    wref = R.allegro_virial(cfg, sd, torch.tensor(pos), torch.tensor(ei), types, sv)
    for got, want in ((e, ref["atomic_energy"].reshape(-1)), (f, ref["forces"]), (w, wref)):
        assert (got - want).abs().max().item() <= 1e-9 * max(1.0, float(want.abs().max()))
