"""Shared graph / model builders and the fp64-oracle criterion of the fast-path tests (test infrastructure)."""
import numpy as np
import torch

from allegro_amd import graph as G
from allegro_amd.nn import HipAllegroModel


def _cfg(embed="bessel", coupling=True, l_max=2, seed=11, avg=9.0, scale_shift=True):
    rce = ({"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8} if embed == "bessel" else
           {"_target_": "allegro.nn.TwoBodySplineScalarEmbed", "num_splines": 8, "spline_span": 6})
    c = dict(type_names=["A", "B"], r_max=3.4, l_max=l_max, num_layers=2, num_scalar_features=64, num_tensor_features=64,
             radial_chemical_embed=rce, radial_chemical_embed_dim=64, scalar_embed_mlp_hidden_layers_width=64,
             allegro_mlp_hidden_layers_width=64, readout_mlp_hidden_layers_width=64, avg_num_neighbors=avg, seed=seed,
             tp_path_channel_coupling=coupling, model_dtype="float32")
    if scale_shift:
        c.update(per_type_energy_scales=[1.3, 0.6], per_type_energy_shifts=[-2.0, 0.25])
    return c


def _ragged(dims=(4, 4, 4), keep=0.88, a=1.8, seed=5, r_cut=3.4):
    """A ragged open cluster: jittered lattice (spacing a) with vacancies -- degrees from a few (corners) up to ~30
    (interior), no unphysically short contacts that would make the fp32 sums ill-conditioned -- plus one isolated
    atom without any edge."""
    rng = np.random.default_rng(seed)
    grid = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1).reshape(-1, 3)
    sel = np.sort(rng.permutation(len(grid))[:int(round(keep * len(grid)))])
    pos = grid[sel] * a + rng.uniform(-0.2, 0.2, size=(len(sel), 3)) + 10.0
    pos = np.concatenate([pos, [[70.0, 70.0, 70.0]]])  # isolated: no edges
    cell = np.eye(3) * 120.0
    ei, shift = G.neighbor_list_pbc(pos, cell, r_cut)
    types = rng.integers(0, 2, size=len(pos))
    return pos, cell, ei, shift, types


def _dense_cluster(n=40, seed=9):
    """40 atoms at 0.7 spacing: every center atom has more than 32 neighbors inside 3.4 (several MFMA tiles per atom)."""
    rng = np.random.default_rng(seed)
    grid = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(3), indexing="ij"), -1).reshape(-1, 3)[:n]
    pos = grid * 0.7 + rng.uniform(-0.05, 0.05, size=(n, 3)) + 20.0
    pos = np.concatenate([pos, [[50.0, 50.0, 50.0]]])  # isolated: no edges
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    keep = ei[0] < 14  # (14 center atoms keep their segments, the rest are neighbors only: emulation time)
    return pos, cell, ei[:, keep], shift[keep], rng.integers(0, 2, size=n + 1)


def _vs_oracle64(cfg, pos, cell, ei, shift, types, lib, dev):
    """HIP fp32 may not be further from the fp64 oracle on the same (upcast) weights than the fp32 CPU oracle is
    (x2 + a small floor) -- the criterion of tests/test_hip_model.py for fp32 sums."""
    from oracle import restatement as R

    n = pos.shape[0]
    m = HipAllegroModel(**cfg).to(dev)
    if lib is not None:
        m._bind_library(lib)
    sv = torch.tensor(shift @ cell, dtype=torch.float32)
    tt = torch.tensor(types)
    g = m.prepare_graph(torch.tensor(ei).to(dev), tt.to(dev), n, sv.to(dev))
    e, f = m.energy_forces(torch.tensor(pos, dtype=torch.float32, device=dev), g)
    e, f = e.cpu(), f.cpu()
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref32 = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=torch.float32), torch.tensor(ei), tt, sv)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = R.allegro_energy_forces(dict(cfg, model_dtype="float64"), sd64, torch.tensor(pos), torch.tensor(ei), tt, sv.double())
    for got, w32, w64 in ((e, ref32["atomic_energy"].reshape(-1), ref64["atomic_energy"].reshape(-1)),
                          (f, ref32["forces"], ref64["forces"])):
        assert torch.isfinite(got).all()
        scale = max(1.0, float(w64.abs().max()))
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu32 = (w32.double() - w64).abs().max().item()
        assert err_hip <= 2.0 * err_cpu32 + 1e-5 * scale, (err_hip, err_cpu32, scale)
    return m



def _assert_launched(m, pos, cell, ei, shift, types, present, absent):
    """The launch list of a step names the kernels that ran."""
    import bench

    g = m.prepare_graph(torch.tensor(ei), torch.tensor(types), pos.shape[0], torch.tensor(shift @ cell, dtype=torch.float32))
    names = [s[0] for s in bench.profile_stages(m, torch.tensor(pos, dtype=torch.float32), g, reps=1)]
    assert all(n in names for n in present) and not any(n in names for n in absent), names
