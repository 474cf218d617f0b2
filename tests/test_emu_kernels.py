"""CPU: the UNMODIFIED HIP kernel sources, compiled against the test-only fiber emulation of the HIP
execution model (tests/emu), checked against the reference's golden vectors.  This exercises the
kernel logic (indexing, CG tables, segments, MFMA fragment mapping, hand-written backward) without a
GPU; the GPU parity tests proper are in test_hip_*.py (-m gpu)."""
import pytest
import torch

from tests.golden_utils import load_contract_cases, load_model_fixture
from tests.hip_utils import emu_lib, fixture_data, model_from_fixture
from allegro_amd.nn import HipContracter


@pytest.mark.parametrize("name,dtype,tol", [
    ("t_coupled", torch.float64, 1e-9), ("t_coupled", torch.float32, 5e-5), ("t_uncoupled", torch.float64, 1e-9),
    ("t_peredge", torch.float64, 1e-9), ("c5_small", torch.float64, 1e-9), ("c1_L1", torch.float32, 5e-5),
    ("t_spline", torch.float64, 1e-9), ("t_spline_peredge", torch.float32, 5e-5),
    ("c2_spline", torch.float32, 5e-5),
    # constructor options beyond the defaults: gelu / mish / None MLPs (general kernels), one env weight per channel
    ("t_acts", torch.float64, 1e-9), ("t_mish", torch.float32, 5e-5), ("t_shared", torch.float64, 1e-9),
    ("c2_shared", torch.float32, 5e-5)])
def test_model_matches_reference_golden(name, dtype, tol):
    fx = load_model_fixture(name, dtype)
    m = model_from_fixture(fx, dtype, emu_lib())
    data, sv = fixture_data(fx, dtype)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    e, f = m.energy_forces(data["pos"], g)
    ref = fx["out"]
    for got, want in ((e, ref["atomic_energy"].reshape(-1)), (f, ref["forces"])):
        scale = max(1.0, float(want.abs().max()))
        assert (got - want).abs().max().item() <= tol * scale


def test_model_unsorted_edges_and_dict_interface():
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    data, sv = fixture_data(fx, torch.float64)
    perm = torch.randperm(data["edge_index"].shape[1], generator=torch.Generator().manual_seed(0))
    g = m.prepare_graph(data["edge_index"][:, perm], data["atom_types"], data["pos"].shape[0], sv[perm])
    e, f = m.energy_forces(data["pos"], g)
    assert (f - fx["out"]["forces"]).abs().max() < 1e-8
    assert (e - fx["out"]["atomic_energy"].reshape(-1)).abs().max() < 1e-8


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
def test_contracter_matches_reference_cases(dtype, tol):
    """Shapes/tolerances of the reference's own kernel test (tests/nn/test_contract_kernels.py:93-134)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        for c in load_contract_cases():
            m = c["meta"]
            mod = HipContracter(m["irreps_in1"], m["irreps_in2"], m["irreps_out"], m["mul"],
                                path_channel_coupling=m["coupling"], scatter_factor=m["scatter_factor"])
            assert mod.w3j_is_ij_diagonal == m["ij_diagonal"] and mod.num_paths == m["num_paths"]
            # the product's own Wigner-3j construction reproduces the reference buffer
            assert (mod.w3j.double() - torch.tensor(c["w3j"])).abs().max() < 1e-6
            mod.load_state_dict({"weights": torch.tensor(c["weights"]).to(dtype), "w3j": torch.tensor(c["w3j"]).to(dtype)})
            mod._bind_library(emu_lib())
            x1 = torch.tensor(c["x1"]).to(dtype).requires_grad_(True)
            x2 = torch.tensor(c["x2"]).to(dtype).requires_grad_(True)
            y = mod(x1, x2, torch.tensor(c["idxs"]), torch.tensor([m["num_atoms"]]))
            g1, g2, gw = torch.autograd.grad(y, [x1, x2, mod.weights], torch.tensor(c["gout"]).to(dtype))
            assert gw.shape == mod.weights.shape
            for got, want in ((y, c["out"]), (g1, c["gx1"]), (g2, c["gx2"])):
                assert (got.double() - torch.tensor(want)).abs().max().item() < tol
            # path-weight gradient (training; the reference's eager autograd, _contract.py:172-177): a sum over all
            # edges (and, uncoupled, channels), hence relative to its own scale
            want = torch.tensor(c["gw"])
            assert (gw.double() - want).abs().max().item() < tol * max(1.0, float(want.abs().max()))
    finally:
        torch.set_default_dtype(old)


def test_moments_packed_path_ragged_degrees_vs_oracle():
    """u=64 / S=64 routes to the moments + packed two-edge kernels; ragged segments (odd degrees, an isolated
    atom) exercise the padded second edge.  Checked against the oracle restatement in fp64."""
    import numpy as np

    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    rng = np.random.default_rng(5)
    pos = rng.uniform(0, 7.5, size=(14, 3))
    pos[13] = [30.0, 30.0, 30.0]  # isolated: degree 0
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    deg = np.bincount(ei[0], minlength=14)
    assert (deg % 2 == 1).any() and deg[13] == 0 and ei.shape[1] > 20
    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=2, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=16, scalar_embed_mlp_hidden_layers_width=32, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=32, avg_num_neighbors=float(deg.mean()), seed=11, model_dtype="float64")
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    types = torch.tensor(rng.integers(0, 2, size=14))
    g = m.prepare_graph(torch.tensor(ei), types, 14, torch.tensor(shift @ cell))
    e, f = m.energy_forces(torch.tensor(pos), g)
    sd = {k[len("func."):]: v.detach() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos), torch.tensor(ei), types, torch.tensor(shift @ cell))
    assert (e - ref["atomic_energy"].reshape(-1)).abs().max() < 1e-9 * max(1.0, float(ref["atomic_energy"].abs().max()))
    assert (f - ref["forces"]).abs().max() < 1e-9 * max(1.0, float(ref["forces"].abs().max()))


def test_moments_path_long_segments_vs_oracle():
    """Degrees above the 64-edge staging chunk of the moments kernels (and odd): every center atom of a dense cluster sees
    all 69 others, so each segment is walked in two staged chunks (64 + 5 edges, padded last pair)."""
    import numpy as np

    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    rng = np.random.default_rng(9)
    grid = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(3), indexing="ij"), -1).reshape(-1, 3)[:70]
    pos = grid * 0.52 + rng.uniform(-0.05, 0.05, size=(70, 3)) + 20.0
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    keep = ei[0] < 12  # (only the first 12 atoms keep their segments: the emulated 64-wide model costs ~15 ms per edge;
    ei, shift = ei[:, keep], shift[keep]  # the rest are neighbors only -- empty segments behind long ones)
    deg = np.bincount(ei[0], minlength=70)
    assert deg[:12].min() == 69 and deg.max() == 69 and deg[12:].max() == 0
    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=2, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=32, scalar_embed_mlp_hidden_layers_width=64, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=69.0, seed=3, model_dtype="float64")
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    types = torch.tensor(rng.integers(0, 2, size=70))
    g = m.prepare_graph(torch.tensor(ei), types, 70, torch.tensor(shift @ cell))
    e, f = m.energy_forces(torch.tensor(pos), g)
    sd = {k[len("func."):]: v.detach() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos), torch.tensor(ei), types, torch.tensor(shift @ cell))
    assert (e - ref["atomic_energy"].reshape(-1)).abs().max() < 1e-9 * max(1.0, float(ref["atomic_energy"].abs().max()))
    assert (f - ref["forces"]).abs().max() < 1e-9 * max(1.0, float(ref["forces"].abs().max()))


@pytest.mark.parametrize("l_max,L,u,S,force,we,slot,extra", [
    (2, 3, 64, 64, False, 32, True, {}), (3, 3, 128, 128, False, 32, False, dict(proj=1)), (3, 3, 128, 128, False, 128, True, dict(proj=1)),
    (2, 2, 128, 64, False, 32, True, dict(radial_chemical_embed_dim=128)), (2, 2, 64, 64, True, 32, True, dict(proj=1)),
    (2, 3, 64, 64, False, 64, True, dict(scalar_embed_mlp_hidden_layers_depth=2, readout_mlp_hidden_layers_depth=2)),
    (2, 3, 64, 64, False, 64, False, dict(allegro_mlp_hidden_layers_depth=2, proj=1))])
def test_operator_path_vs_oracle(l_max, L, u, S, force, we, slot, extra, monkeypatch):
    """Per-atom operator form of the tensor-product track (aa_tp_op.hip): 3-layer stacks, two 64-channel slices,
    and (forced) the 2-layer case the tuned kernels normally take.  fp64 against the oracle restatement.  `slot`: whether the
    plan takes the slot form of the linear layers (`proj`: and the env projections as batched launches; output layers of scalar_embed_mlp / the latent MLPs folded into their
    consumers, reverse pass per dense-net slot: scalar_embed_mlp and latent hidden widths equal to S, one hidden latent layer)."""
    import numpy as np

    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    if force:
        monkeypatch.setenv("AA_TP_OP", "1")
    extra = dict(extra)
    if extra.pop("proj", 0):  # the env projections as batched linear-layer launches (default from 4096 atoms on)
        monkeypatch.setenv("AA_OP_PROJ", "1")
    rng = np.random.default_rng(5)
    n = 10
    pos = rng.uniform(0, 6.5, size=(n, 3))
    pos[n - 1] = [30.0, 30.0, 30.0]
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    deg = np.bincount(ei[0], minlength=n)
    assert deg[n - 1] == 0 and ei.shape[1] >= 10
    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=l_max, num_layers=L, num_scalar_features=S, num_tensor_features=u,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               scalar_embed_mlp_hidden_layers_width=we, allegro_mlp_hidden_layers_width=S,
               readout_mlp_hidden_layers_width=32, avg_num_neighbors=float(deg.mean()), seed=11, model_dtype="float64",
               **{"radial_chemical_embed_dim": 16, **extra})  # (embed dim 128: the 16-lanes-per-edge form of the geometry reverse)
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    d = m.describe_plan()
    assert d["operator_path"] and d["slot_form"] == slot, d
    types = torch.tensor(rng.integers(0, 2, size=n))
    g = m.prepare_graph(torch.tensor(ei), types, n, torch.tensor(shift @ cell))
    e, f = m.energy_forces(torch.tensor(pos), g)
    sd = {k[len("func."):]: v.detach() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos), torch.tensor(ei), types, torch.tensor(shift @ cell))
    assert (e - ref["atomic_energy"].reshape(-1)).abs().max() < 1e-9 * max(1.0, float(ref["atomic_energy"].abs().max()))
    assert (f - ref["forces"]).abs().max() < 1e-9 * max(1.0, float(ref["forces"].abs().max()))


@pytest.mark.parametrize("dt,tol", [("float64", 1e-11), ("float32", 1e-4)])
def test_slot_form_matches_the_unfolded_pipeline(dt, tol, monkeypatch):
    """Same model, same graph: the slot form (default) against aa_plan_options.no_slot_form -- two different launch sequences and
    weight sets (folded at pack time vs the reference's own matrices) that must agree to rounding; energies, forces, virial."""
    import numpy as np

    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    rng = np.random.default_rng(8)
    n = 12
    pos = rng.uniform(0, 6.0, size=(n, 3))
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    deg = np.bincount(ei[0], minlength=n)
    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=3, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=16, scalar_embed_mlp_hidden_layers_width=64, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=float(deg.mean()), seed=3, model_dtype=dt,
               per_type_energy_scales=[1.3, 0.6], per_type_energy_shifts=[-2.0, 0.25])
    tdt = getattr(torch, dt)
    types = torch.tensor(rng.integers(0, 2, size=n))
    out = []
    for unfolded, proj in ((False, "1"), (True, "0")):  # (slot form + batched env projections) vs (neither)
        monkeypatch.setenv("AA_OP_PROJ", proj)
        if unfolded:
            monkeypatch.setenv("AA_NO_SLOT_FORM", "1")
            monkeypatch.setenv("AA_OP_RECOMPUTE_BVECS", "1")  # (and the layer-0 reverse recomputing the per-atom vectors B_l)
            monkeypatch.setenv("AA_READOUT_TWO_PASS", "1")    # (and d E / d readout hidden by its own kernel)
        m = HipAllegroModel(**cfg)
        m._bind_library(emu_lib())
        assert m.describe_plan()["slot_form"] == (not unfolded)
        g = m.prepare_graph(torch.tensor(ei), types, n, torch.tensor(shift @ cell, dtype=tdt))
        e, f = m.energy_forces(torch.tensor(pos, dtype=tdt), g)
        out.append((e.double(), f.double(), m.virial(g).double()))
    for a, b in zip(*out):
        assert torch.isfinite(a).all() and (a - b).abs().max().item() <= tol * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_graph_without_edges_gives_shifts_and_zero_forces(dt):
    """Empty edge list (every atom isolated): E_i = per-type shift, F = 0; no kernel may touch an edge."""
    from allegro_amd.nn import HipAllegroModel

    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=2, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=64, scalar_embed_mlp_hidden_layers_width=64, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=10.0, seed=11, model_dtype=dt,
               per_type_energy_shifts=[1.5, -2.0])
    tdt = torch.float64 if dt == "float64" else torch.float32
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    pos = torch.tensor([[0.0, 0, 0], [10, 0, 0], [0, 10, 0]], dtype=tdt)
    g = m.prepare_graph(torch.zeros((2, 0), dtype=torch.long), torch.tensor([0, 1, 0]), 3, None)
    e, f = m.energy_forces(pos, g)
    assert e.tolist() == [1.5, -2.0, 1.5] and float(f.abs().max()) == 0.0


@pytest.mark.parametrize("name,dtype,tol", [("t_coupled", torch.float64, 1e-9), ("c5_small", torch.float64, 1e-9),
                                            ("c1_L1", torch.float32, 5e-5)])
def test_virial_matches_oracle_strain_derivative(name, dtype, tol):
    """aa_model_virial: W = sum_e dE/dr_e (x) r_e  ==  dE/d(strain) of the oracle (autograd through a strained copy
    of positions + periodic shifts); symmetric because the energy is rotation invariant."""
    from oracle import restatement as R

    fx = load_model_fixture(name, dtype)
    m = model_from_fixture(fx, dtype, emu_lib())
    data, sv = fixture_data(fx, dtype)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    m.energy_forces(data["pos"], g)
    w = m.virial(g)
    cfg = dict(fx["cfg"])
    cfg["model_dtype"] = "float64"
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in fx["sd"].items()}
    ref = R.allegro_virial(cfg, sd, fx["pos"].double(), fx["edge_index"], fx["types"],
                           None if fx["shift_vec"] is None else fx["shift_vec"].double())
    scale = max(1.0, float(ref.abs().max()))
    assert (w.double() - ref).abs().max().item() <= tol * scale
    assert (w - w.T).abs().max().item() <= 10 * tol * scale


def test_f64_gemm_column_tile_loop_matches_2d_grid(monkeypatch):
    """fp64 MFMA GEMM: one workgroup per 128-row tile walking all column tiles (large systems: A comes from HBM once)
    must equal the 2-D grid launch bit for bit (same arithmetic order)."""
    fx = load_model_fixture("c5_small", torch.float64)
    data, sv = fixture_data(fx, torch.float64)
    outs = []
    for mode in ("0", "2"):
        monkeypatch.setenv("AA_F64_NLOOP", mode)
        m = model_from_fixture(fx, torch.float64, emu_lib())
        g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
        outs.append(m.energy_forces(data["pos"], g))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want = fx["out"]["forces"]
    assert (outs[1][1] - want).abs().max().item() <= 1e-9 * max(1.0, float(want.abs().max()))


def _tc_model():
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    data, sv = fixture_data(fx, torch.float64)
    return fx, m, data, sv


def test_symmetries_of_the_hip_path():
    """The invariances the reference's own model tests assert (BaseEnergyModelTests via tests/model/test_allegro.py):
    permutation equivariance, translation invariance, parity (E even, F odd) -- through the kernels themselves."""
    fx, m, data, sv = _tc_model()
    pos, ei, ty = data["pos"], data["edge_index"], data["atom_types"]
    n = pos.shape[0]
    e0, f0 = m.energy_forces(pos, m.prepare_graph(ei, ty, n, sv))
    # permutation of the atoms (edge list relabelled accordingly; it becomes unsorted -> sorted internally)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(4))
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    e1, f1 = m.energy_forces(pos[perm], m.prepare_graph(inv[ei], ty[perm], n, sv))
    assert (e1 - e0[perm]).abs().max() < 1e-10 and (f1 - f0[perm]).abs().max() < 1e-10
    # rigid translation
    e2, f2 = m.energy_forces(pos + torch.tensor([0.37, -1.2, 2.5], dtype=pos.dtype), m.prepare_graph(ei, ty, n, sv))
    assert (e2 - e0).abs().max() < 1e-10 and (f2 - f0).abs().max() < 1e-10
    # inversion through the origin: parity=True model, scalar energy
    e3, f3 = m.energy_forces(-pos, m.prepare_graph(ei, ty, n, -sv))
    assert (e3 - e0).abs().max() < 1e-10 and (f3 + f0).abs().max() < 1e-10


def test_strict_locality_of_the_hip_path():
    """tests/model/test_allegro.py:68-70 (`strict_locality`): E_i depends only on atoms inside i's cutoff sphere, and
    no force acts between atoms that are not neighbors of a common center ... checked by displacing one atom: only
    the energies of the centers that list it change, and dE_i/dx_j is zero for every other center."""
    fx, m, data, sv = _tc_model()
    pos, ei, ty = data["pos"], data["edge_index"], data["atom_types"]
    n = pos.shape[0]
    g = m.prepare_graph(ei, ty, n, sv)
    e0, _ = m.energy_forces(pos, g)
    j = int(torch.bincount(ei[1], minlength=n).argmin())  # the atom with the fewest centers listing it
    moved = pos.clone()
    moved[j] += torch.tensor([0.01, -0.02, 0.015], dtype=pos.dtype)  # small: the neighbor sets stay the same
    e1, _ = m.energy_forces(moved, g)
    touched = torch.zeros(n, dtype=torch.bool)
    touched[ei[0][ei[1] == j]] = True   # centers that have j as a neighbor
    touched[j] = True                   # and j itself (all its own edge vectors move)
    changed = (e1 - e0).abs() > 1e-13
    assert not bool((changed & ~touched).any())
    assert bool(changed[touched].any())


@pytest.mark.parametrize("n,box,L,l_max,dt", [(2, 2.0, 2, 2, "float64"), (3, 2.5, 2, 2, "float64"), (9, 6.0, 2, 2, "float64"),
                                              (7, 5.0, 3, 2, "float64"), (7, 5.0, 2, 3, "float64"),
                                              (2, 2.0, 2, 2, "float32"), (9, 6.0, 2, 2, "float32")])
def test_tiny_graphs_on_the_fast_paths_vs_oracle(n, box, L, l_max, dt):
    """Fewer edges than one 32-row GEMM tile / one wave (E = 2 ... 28): tile masking, single-edge segments and
    several periodic images of the same pair, on the 64-wide fast paths (moments + chains, operator path for L = 3)."""
    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel
    import numpy as np

    rng = np.random.default_rng(100 + n + L + l_max)
    rc = 3.0
    pos = rng.uniform(0, box, size=(n, 3))
    cell = np.eye(3) * (2 * rc + 0.5)
    ei, shift = G.neighbor_list_pbc(pos, cell, rc)
    assert 0 < ei.shape[1] < 32 or n >= 7
    deg = np.bincount(ei[0], minlength=n)
    cfg = dict(type_names=["A", "B"], r_max=rc, l_max=l_max, num_layers=L, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               scalar_embed_mlp_hidden_layers_width=64, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=max(1.0, float(deg.mean())), seed=3,
               model_dtype=dt)
    dtype, tol = (torch.float64, 1e-9) if dt == "float64" else (torch.float32, 5e-5)
    m = HipAllegroModel(**cfg)
    m._bind_library(emu_lib())
    types = torch.tensor(rng.integers(0, 2, size=n))
    sv = torch.tensor(shift @ cell, dtype=dtype)
    e, f = m.energy_forces(torch.tensor(pos, dtype=dtype), m.prepare_graph(torch.tensor(ei), types, n, sv))
    sd = {k[len("func."):]: v.detach() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=dtype), torch.tensor(ei), types, sv)
    for got, want in ((e, ref["atomic_energy"].reshape(-1)), (f, ref["forces"])):
        assert (got - want).abs().max().item() <= tol * max(1.0, float(want.abs().max()))


def test_bessel_sinc_convention_is_recognised_from_the_stored_roots():
    """nequip's BesselEdgeLengthEncoding exists in two published forms (roots n*pi with sin(w x)/x, or roots n with
    sinc(x w) w): a checkpoint of either kind must evaluate correctly.  The packed weights switch on the stored
    `bessel_weights`; checked against the oracle, which applies the same rule with torch.sinc."""
    from oracle import restatement as R

    fx = load_model_fixture("t_coupled", torch.float64)
    sd = dict(fx["sd"])
    key = "radial_chemical_embed.bessel_encode.bessel_weights"
    n = sd[key].numel()
    sd[key] = torch.arange(1, n + 1, dtype=torch.float64).reshape(sd[key].shape)
    fx2 = dict(fx, sd=sd)
    m = model_from_fixture(fx2, torch.float64, emu_lib())
    data, sv = fixture_data(fx2, torch.float64)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    e, f = m.energy_forces(data["pos"], g)
    cfg = dict(fx["cfg"], model_dtype="float64")
    ref = R.allegro_energy_forces(cfg, sd, fx["pos"], fx["edge_index"], fx["types"], fx["shift_vec"])
    assert (e - ref["atomic_energy"].reshape(-1)).abs().max() < 1e-9
    assert (f - ref["forces"]).abs().max() < 1e-9
    # and it IS a different function of the same weights than the n*pi form (1/pi prefactor)
    assert (f - fx["out"]["forces"]).abs().max() > 1e-3


@pytest.mark.parametrize("u,coupling,individual,layers", [(32, True, True, 3), (48, True, False, 2)])  # (u = 16: the constructor-defaults case below; u = 96 -> 128: the GPU test)
def test_channel_counts_off_the_multiples_of_64_run_zero_padded(u, coupling, individual, layers, monkeypatch):
    """Channel padding (aa_model_plan_create): a stack whose tensor-channel count is not a multiple of 64 is evaluated as
    the next multiple-of-64 stack whose extra channels have zero weights -- same energies and forces as the narrow model
    (fp64 oracle criterion), on the moments / operator / chain kernels (launch list); AA_NO_PAD=1 keeps the narrow
    kernels and agrees."""
    import numpy as np

    import bench
    from tests.fastpath_utils import _cfg, _ragged
    from tests.fastpath_utils import _vs_oracle64

    pos, cell, ei, shift, types = _ragged(dims=(3, 3, 2), keep=0.9, seed=4)
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    cfg = _cfg("bessel", coupling, avg=float(deg.mean()))
    cfg.update(num_tensor_features=u, weight_individual_irreps=individual, num_layers=layers)
    fast = "tp_mom_fwd_first" if layers == 2 and u < 64 else "tp_op_fwd"  # 3 layers / 128 channels: the per-atom operator kernels
    out = {}
    for no_pad in ("0", "1"):
        monkeypatch.setenv("AA_NO_PAD", no_pad)
        if no_pad == "0":
            m = _vs_oracle64(cfg, pos, cell, ei, shift, types, emu_lib(), torch.device("cpu"))
            sd = m.state_dict()
        else:  # (the narrow kernels have their own oracle tests; here: same weights, same answer)
            from allegro_amd.nn import HipAllegroModel

            m = HipAllegroModel(**cfg)
            m.load_state_dict(sd)
            m._bind_library(emu_lib())
        g = m.prepare_graph(torch.tensor(ei), torch.tensor(types), pos.shape[0], torch.tensor(shift @ cell, dtype=torch.float32))
        p32 = torch.tensor(pos, dtype=torch.float32)
        names = [s[0] for s in bench.profile_stages(m, p32, g, reps=1)]
        # (padded 2-layer stacks land on the 64-channel kernels: by default the fused forward, else the moments kernels)
        assert (fast in names or "fused_fwd" in names) == (no_pad == "0"), names
        out[no_pad] = m.energy_forces(p32, g)
    assert (out["0"][1] - out["1"][1]).abs().max().item() < 2e-5 * max(1.0, float(out["1"][1].abs().max()))


@pytest.mark.parametrize("widths", [dict(readout_mlp_hidden_layers_width=32, num_tensor_features=16),  # the reference's constructor defaults
                                    dict(scalar_embed_mlp_hidden_layers_width=32, allegro_mlp_hidden_layers_width=48, readout_mlp_hidden_layers_width=8)])
def test_narrow_hidden_layers_run_zero_padded_on_the_fused_chains(widths, monkeypatch):
    """Hidden-width padding (aa_model_plan_create): single hidden layers narrower than 64 -- e.g. the constructor default
    readout width 32 (allegro_models.py:137) -- are zero-padded to 64 (silu(0) = 0, no biases), which puts the stack on the
    fused linear-layer chains; results are those of the narrow model (fp64 oracle criterion; AA_NO_PAD=1 agrees)."""
    import numpy as np

    import bench
    from tests.fastpath_utils import _cfg, _ragged
    from tests.fastpath_utils import _vs_oracle64

    pos, cell, ei, shift, types = _ragged(dims=(3, 3, 2), keep=0.9, seed=4)
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    cfg = _cfg("bessel", True, avg=float(deg.mean()))
    cfg.update(widths)
    out = {}
    for no_pad in ("0", "1"):
        monkeypatch.setenv("AA_NO_PAD", no_pad)
        if no_pad == "0":
            m = _vs_oracle64(cfg, pos, cell, ei, shift, types, emu_lib(), torch.device("cpu"))
            sd = m.state_dict()
        else:  # (the narrow kernels have their own oracle tests; here: same weights, same answer)
            from allegro_amd.nn import HipAllegroModel

            m = HipAllegroModel(**cfg)
            m.load_state_dict(sd)
            m._bind_library(emu_lib())
        g = m.prepare_graph(torch.tensor(ei), torch.tensor(types), pos.shape[0], torch.tensor(shift @ cell, dtype=torch.float32))
        p32 = torch.tensor(pos, dtype=torch.float32)
        names = [s[0] for s in bench.profile_stages(m, p32, g, reps=1)]
        assert any(n.startswith("gc_") for n in names) == (no_pad == "0"), names
        out[no_pad] = m.energy_forces(p32, g)
    assert (out["0"][1] - out["1"][1]).abs().max().item() < 2e-5 * max(1.0, float(out["1"][1].abs().max()))


@pytest.mark.parametrize("dt,S,lat,tol", [("float32", 128, 128, 2e-4), ("float64", 64, 64, 1e-10), ("float64", 128, 128, 1e-10),
                                          ("float32", 64, 128, 2e-4)])
def test_two_layer_stacks_off_the_chain_shapes_take_the_operator_path(dt, S, lat, tol, monkeypatch):
    """Round 5 (profiles/r05_v3 / v7 / v8 shape maps): 2-layer u = 64 stacks the fused chains do not cover -- fp64, or S / MLP widths
    of 128 -- ran the 2-layer moments kernels + single linear layers; the operator kernels (+ slot form) are faster there (fp32
    u 64 / S 128: 7.41 -> 3.96 ms at C3; fp64 u = S = 64: 4.89 -> 4.14 ms) and are now selected, except where the slot form does not
    apply in fp32 (S 64 with 128-wide latents: a wash, keeps the moments kernels).  Both selections against the oracle."""
    import numpy as np

    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel
    from oracle import restatement as R

    rng = np.random.default_rng(21)
    n = 9
    pos = rng.uniform(0, 5.5, size=(n, 3))
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    deg = np.bincount(ei[0], minlength=n)
    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=2, num_scalar_features=S, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8}, radial_chemical_embed_dim=S,
               scalar_embed_mlp_hidden_layers_width=S, allegro_mlp_hidden_layers_width=lat, readout_mlp_hidden_layers_width=lat,
               avg_num_neighbors=float(deg.mean()), seed=5, model_dtype=dt)
    tdt = getattr(torch, dt)
    types = torch.tensor(rng.integers(0, 2, size=n))
    expect_op = not (dt == "float32" and S == 64)
    outs = []
    for prefer_moments in (False, True):
        if prefer_moments:
            monkeypatch.setenv("AA_TP_PREFER_MOM", "1")
        m = HipAllegroModel(**cfg)
        m._bind_library(emu_lib())
        d = m.describe_plan()
        assert d["operator_path"] == (expect_op and not prefer_moments), (d, prefer_moments)
        if d["operator_path"]:
            assert d["slot_form"] == (lat == S)
        g = m.prepare_graph(torch.tensor(ei), types, n, torch.tensor(shift @ cell, dtype=tdt))
        e, f = m.energy_forces(torch.tensor(pos, dtype=tdt), g)
        outs.append((e.double(), f.double()))
        if not prefer_moments:
            sd = {k[len("func."):]: v.detach() for k, v in m.state_dict().items()}
            ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=tdt), torch.tensor(ei), types, torch.tensor(shift @ cell, dtype=tdt))
            assert (e - ref["atomic_energy"].reshape(-1)).abs().max() <= tol * max(1.0, float(ref["atomic_energy"].abs().max()))
            assert (f - ref["forces"]).abs().max() <= tol * max(1.0, float(ref["forces"].abs().max()))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= tol * max(1.0, float(b.abs().max()))
