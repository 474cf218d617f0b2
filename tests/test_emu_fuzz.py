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


@pytest.mark.parametrize("seed", [1, 3, 8, 13, 20, 22])
def test_random_fp32_fast_path_configuration_is_as_accurate_as_the_fp32_oracle(seed):
    """The specialised fp32 paths (64/128-wide: moments or operator TP kernels, bf16x3 GEMM chains or single layers),
    random species / l_max / layers / embedding: judged against the fp64 oracle on the same (upcast) weights -- the HIP
    result may not be further from it than the fp32 CPU oracle is (x2 + 1e-5 of the force scale)."""
    from oracle import restatement as R

    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(6, 30))
    rc = float(rng.uniform(2.8, 3.6))
    box = float(rng.uniform(2 * rc + 0.2, 10.0))
    k = int(np.ceil(n ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
    pos = (grid + 0.5 + rng.uniform(-0.25, 0.25, (n, 3))) * (box / k)  # jittered lattice: no unphysically close pairs
    cell = np.eye(3) * box
    ei, shift = G.neighbor_list_pbc(pos, cell, rc)
    assert ei.shape[1] > 0
    T, l_max, L = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.choice([2, 2, 3]))
    wide = bool(rng.integers(0, 3) == 0)
    u = 128 if wide else 64
    S = 128 if (wide and rng.integers(0, 2)) else 64
    H = 128 if (wide and rng.integers(0, 2)) else 64
    spline = bool(rng.integers(0, 2))
    rce = ({"_target_": "allegro.nn.TwoBodySplineScalarEmbed", "num_splines": 8, "spline_span": 6} if spline else
           {"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8})
    cfg = dict(type_names=["A", "B", "C"][:T], r_max=rc, l_max=l_max, num_layers=L, num_scalar_features=S,
               num_tensor_features=u, radial_chemical_embed=rce,
               radial_chemical_embed_dim=(None if rng.integers(0, 2) else 32), scalar_embed_mlp_hidden_layers_width=64,
               allegro_mlp_hidden_layers_width=H, readout_mlp_hidden_layers_width=int(rng.choice([32, 64])),
               tp_path_channel_coupling=bool(rng.integers(0, 4) > 0), avg_num_neighbors=float(max(1, ei.shape[1] / n)),
               seed=int(seed), model_dtype="float32")
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    types = torch.tensor(rng.integers(0, T, size=n))
    sv = torch.tensor(shift @ cell, dtype=torch.float32)
    e, f = m.energy_forces(torch.tensor(pos, dtype=torch.float32), m.prepare_graph(torch.tensor(ei), types, n, sv))
    sd = {k_[len("func."):]: v.detach() for k_, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=torch.float32), torch.tensor(ei), types, sv)
    sd64 = {k_: (v.double() if v.is_floating_point() else v) for k_, v in sd.items()}
    ref64 = R.allegro_energy_forces(dict(cfg, model_dtype="float64"), sd64, torch.tensor(pos), torch.tensor(ei), types,
                                    sv.double())
    for got, w32, w64 in ((e, ref["atomic_energy"].reshape(-1), ref64["atomic_energy"].reshape(-1)),
                          (f, ref["forces"], ref64["forces"])):
        assert torch.isfinite(got).all()
        scale = max(1.0, float(w64.abs().max()))
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu32 = (w32.double() - w64).abs().max().item()
        assert err_hip <= 2.0 * err_cpu32 + 1e-5 * scale, (err_hip, err_cpu32, scale)


@pytest.mark.parametrize("seed", [0, 3, 7, 11, 19, 23, 31, 42])
def test_random_irreps_contracter_vs_oracle(seed):
    """The operator seam (HipContracter = Contracter.forward, _contract.py:185-251) on random irreps of both parities
    up to l = 3, random multiplicity, both weight modes, unsorted segment indices: output and both input gradients."""
    from oracle import restatement as R
    from allegro_amd.nn import HipContracter

    rng = np.random.default_rng(seed)
    allir = [f"{l}{p}" for l in range(4) for p in "eo"]
    for _ in range(20):  # redraw until the random irreps admit at least one path
        def pick(kmax):
            return " + ".join(rng.choice(allir, size=int(rng.integers(1, kmax + 1)), replace=False))

        i1, i2, io = pick(4), pick(3), pick(4)
        mul, cpl = int(rng.choice([1, 3, 8, 64])), bool(rng.integers(0, 2))
        torch.manual_seed(seed)
        try:
            c = HipContracter(i1, i2, io, mul=mul, path_channel_coupling=cpl, scatter_factor=float(rng.uniform(0.2, 1.5))).double()
            break
        except Exception:
            continue
    c._bind_library(emu_lib())
    E, N = int(rng.integers(1, 40)), int(rng.integers(1, 9))
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(E, mul, c.base_dim1, dtype=torch.float64, generator=g, requires_grad=True)
    x2 = torch.randn(E, mul, c.base_dim2, dtype=torch.float64, generator=g, requires_grad=True)
    idxs = torch.randint(0, N, (E,), generator=g)
    y = c(x1, x2, idxs, N)
    gy = torch.randn(y.shape, dtype=torch.float64, generator=g)
    g1, g2 = torch.autograd.grad(y, [x1, x2], gy)
    yr = R.contracter_forward(x1, x2, idxs, N, c.weights.detach(), c.w3j, cpl, c.scatter_factor)
    r1, r2 = torch.autograd.grad(yr, [x1, x2], gy)
    for got, want in ((y, yr), (g1, r1), (g2, r2)):
        assert (got - want).abs().max().item() < 1e-10


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_slot_form_configuration_vs_oracle(seed, monkeypatch):
    """The same sweep over the shapes that take the slot form of the linear layers on the operator kernels (S = latent width
    = scalar_embed_mlp width in {64, 128}, 2-3 layers, channel counts that pad to 64 or 128, scalar_embed_mlp / readout depths
    1-2, Bessel or spline bases), with the env projections inside the per-atom kernels or as batched launches, against the
    oracle in fp64: energies, forces, strain derivative; unsorted edge lists, atoms without edges."""
    from oracle import restatement as R

    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(5, 13))
    rc = float(rng.uniform(2.6, 3.4))
    pos = rng.uniform(0, 5.5, (n, 3))
    pos[n - 1] = [40.0, 40.0, 40.0]  # no edges
    cell = np.eye(3) * 80.0
    ei, shift = G.neighbor_list_pbc(pos, cell, rc)
    assert ei.shape[1] > 0
    T, l_max, L = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(2, 4))
    S = int(rng.choice([64, 64, 128]))
    u = int(rng.choice([32, 64, 96, 128]))
    spline, B = bool(rng.integers(0, 2)), int(rng.choice([4, 8]))
    rce = ({"_target_": "allegro.nn.TwoBodySplineScalarEmbed", "num_splines": B, "spline_span": int(rng.integers(1, B + 1))}
           if spline else {"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": B})
    cfg = dict(type_names=["A", "B", "C"][:T], r_max=rc, l_max=l_max, num_layers=L, num_scalar_features=S,
               num_tensor_features=u, radial_chemical_embed=rce, radial_chemical_embed_dim=int(rng.choice([16, 48])),
               scalar_embed_mlp_hidden_layers_depth=int(rng.integers(1, 3)), scalar_embed_mlp_hidden_layers_width=S,
               allegro_mlp_hidden_layers_depth=1, allegro_mlp_hidden_layers_width=S,
               readout_mlp_hidden_layers_depth=int(rng.integers(1, 3)), readout_mlp_hidden_layers_width=int(rng.choice([16, 64, 128])),
               tp_path_channel_coupling=bool(rng.integers(0, 2)), avg_num_neighbors=float(max(1, ei.shape[1] / n)),
               seed=int(seed),
               per_type_energy_scales=[float(x) for x in rng.uniform(0.5, 2, T)] if rng.integers(0, 2) else None,
               per_type_energy_shifts=[float(x) for x in rng.uniform(-1, 1, T)] if rng.integers(0, 2) else None,
               model_dtype="float64")
    monkeypatch.setenv("AA_OP_PROJ", "1" if rng.integers(0, 2) else "0")
    monkeypatch.setenv("AA_TP_OP", "1")  # (operator kernels also where the tuned 2-layer u = 64 kernels would apply)
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    d = m.describe_plan()
    assert d["operator_path"] and d["slot_form"], (cfg, d)
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
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() <= 1e-9 * max(1.0, float(want.abs().max()))
