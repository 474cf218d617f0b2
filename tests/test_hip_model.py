"""GPU parity tests proper: the HIP path (through the C ABI) vs the reference's golden vectors, vs the
oracle restatement on the same seeded inputs, and size-independent properties at BASELINE sizes."""
import math

import numpy as np
import pytest
import torch

from tests.golden_utils import MODEL_FIXTURES, load_model_fixture
from tests.hip_utils import fixture_data, model_from_fixture

pytestmark = pytest.mark.gpu

# the reference's own whole-model tolerances (tests/model/test_allegro.py:72-74), relative to output scale
TOL = {torch.float64: 1e-9, torch.float32: 5e-5}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _both_forwards(forward_mode):
    """Every test of this module runs through the default (automatic: fused per-atom-tile forward wherever the graph
    allows) AND through the staged pipeline (tests/conftest.py: forward_mode)."""
    return forward_mode


def _run(fx, dtype, dev):
    m = model_from_fixture(fx, dtype, device=dev)
    data, sv = fixture_data(fx, dtype, dev)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    e, f = m.energy_forces(data["pos"], g)
    torch.cuda.synchronize()
    return m, g, e.cpu(), f.cpu()


def _launches(m, g, pos):
    """Kernel symbols of one step (aa_model_energy_forces_profiled)."""
    import bench

    return [st[0] for st in bench.profile_stages(m, pos, g, reps=1)]


@pytest.mark.parametrize("name", MODEL_FIXTURES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_energy_forces_match_reference_golden(name, dtype, dev):
    fx = load_model_fixture(name, dtype)
    _, _, e, f = _run(fx, dtype, dev)
    ref = fx["out"]
    for got, want, what in ((e, ref["atomic_energy"].reshape(-1), "E_i"), (f, ref["forces"], "F")):
        scale = max(1.0, float(want.abs().max()))
        err = (got - want).abs().max().item()
        assert err <= TOL[dtype] * scale, f"{name} {what}: {err:.3e} > {TOL[dtype] * scale:.3e}"


def test_c2_forces_within_1e4_eV_per_A(dev):
    """north_star: forces within 1e-4 eV/A of the reference (fp32, BASELINE config 1)."""
    fx = load_model_fixture("c2", torch.float32)
    _, _, e, f = _run(fx, torch.float32, dev)
    assert (f - fx["out"]["forces"]).abs().max().item() < 1e-4
    # also against the fp64 reference
    fx64 = load_model_fixture("c2", torch.float64)
    assert (f.double() - fx64["out"]["forces"]).abs().max().item() < 1e-4


@pytest.mark.parametrize("name,dtype", [("c2", torch.float32), ("t_coupled", torch.float64), ("c5_small", torch.float64)])
def test_intermediates_match_oracle(name, dtype, dev, monkeypatch):
    from oracle import restatement as R

    monkeypatch.setenv("AA_EMBED_NOFUSE", "1")  # the fused path never materialises the two-body embedding tap

    fx = load_model_fixture(name, dtype)
    # taps are a contract of `enable_debug_taps` (aa_model_plan_enable_taps): the fused forward keeps the intermediates on chip --
    # and since round 4 never forms the embedding at all (its output layer is folded into the consumers) -- so the step that is
    # tapped runs the staged pipeline
    m = model_from_fixture(fx, dtype, device=dev)
    m.enable_debug_taps(True)
    data, sv = fixture_data(fx, dtype, dev)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    e, f = m.energy_forces(data["pos"], g)
    ref = fx["out"]
    assert (f.cpu() - ref["forces"]).abs().max().item() <= TOL[dtype] * max(1.0, float(ref["forces"].abs().max()))
    _, inter = R.allegro_energy(fx["cfg"], fx["sd"], fx["pos"], fx["edge_index"], fx["types"], fx["shift_vec"],
                                return_intermediates=True)
    for tap in ("edge_attrs", "emb0", "edge_embedding", "edge_features"):
        got = m.debug_tap(tap, g).cpu()
        want = inter[tap]
        scale = max(1.0, float(want.abs().max()))
        assert (got - want).abs().max().item() <= TOL[dtype] * scale, tap


def test_unsorted_edge_list_is_sorted_internally(dev):
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, device=dev)
    data, sv = fixture_data(fx, torch.float64, dev)
    perm = torch.randperm(data["edge_index"].shape[1], generator=torch.Generator().manual_seed(1)).to(dev)
    g = m.prepare_graph(data["edge_index"][:, perm], data["atom_types"], data["pos"].shape[0], sv[perm])
    e, f = m.energy_forces(data["pos"], g)
    assert (f.cpu() - fx["out"]["forces"]).abs().max() < 1e-8


def test_ghost_layout_matches_pbc_layout(dev):
    """pair_allegro tensor contract (allegro/_compile.py:28-63): ghosts appended, no cell; ghost rows carry
    the force contributions LAMMPS would reverse-communicate -- folded back they equal the PBC forces."""
    from allegro_amd import graph as G

    fx = load_model_fixture("c2", torch.float64)
    m = model_from_fixture(fx, torch.float64, device=dev)
    g = G.make_si_graph(2)
    assert np.allclose(g.pos, fx["pos"].numpy())
    gg = G.to_ghost_layout(g)
    pos = torch.tensor(gg.pos, device=dev)
    pg = m.prepare_graph(torch.tensor(gg.edge_index, device=dev), torch.tensor(gg.types, device=dev), gg.num_atoms)
    e, f = m.energy_forces(pos, pg)
    e, f = e.cpu(), f.cpu()
    n = gg.n_local
    outside = np.abs(g.cell_shift).sum(-1) != 0
    src = torch.tensor(g.edge_index[1][outside])
    folded = f[:n].clone().index_add_(0, src, f[n:])
    assert (folded - fx["out"]["forces"]).abs().max() < 1e-9
    assert (e[:n] - fx["out"]["atomic_energy"].reshape(-1)).abs().max() < 1e-9


def test_finite_difference_forces_fp64(dev):
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, device=dev)
    data, sv = fixture_data(fx, torch.float64, dev)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    pos = data["pos"]
    _, f = m.energy_forces(pos, g)
    h = 1e-5
    for (a, c) in [(0, 0), (3, 1), (11, 2), (20, 0)]:
        p1, p2 = pos.clone(), pos.clone()
        p1[a, c] += h
        p2[a, c] -= h
        e1, _ = m.energy_forces(p1, g, with_forces=False)
        e1 = e1.sum().item()
        e2, _ = m.energy_forces(p2, g, with_forces=False)
        e2 = e2.sum().item()
        fd = -(e1 - e2) / (2 * h)
        assert abs(fd - f[a, c].item()) < 1e-6 * max(1.0, abs(fd))


# ---------------------------------------------------------------------------------- BASELINE sizes
def _big(dev, cells=11):
    import bench
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    g = G.make_si_graph(cells)
    cfg = bench.si_model_cfg(g.num_edges / g.num_atoms)
    cfg["model_dtype"] = "float32"
    m = HipAllegroModel(**cfg).to(dev)
    return g, cfg, m


def test_c3_properties_and_oracle_sample(dev):
    """BASELINE config 2 (10 648 atoms / 298 144 edges): size-independent properties + oracle on a sample."""
    from oracle import restatement as R
    from allegro_amd import graph as G

    g, cfg, m = _big(dev, 11)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    ei = torch.tensor(g.edge_index, device=dev)
    types = torch.tensor(g.types, device=dev)
    sv = torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev)
    pg = m.prepare_graph(ei, types, g.num_atoms, sv)
    e, f = m.energy_forces(pos, pg)
    e, f = e.cpu().double(), f.cpu().double()
    fscale = f.abs().max().item()
    assert torch.isfinite(e).all() and torch.isfinite(f).all() and fscale > 1e-3
    # Newton's third law under PBC: net force vanishes
    assert f.sum(0).abs().max().item() < 1e-3 * fscale * math.sqrt(g.num_atoms)
    # translation invariance
    e2, f2 = m.energy_forces(pos + torch.tensor([0.3, -1.1, 2.7], device=dev), pg)
    assert (e2.cpu().double() - e).abs().max().item() < 5e-4 * max(1.0, e.abs().max().item())
    assert (f2.cpu().double() - f).abs().max().item() < 5e-4 * fscale
    # atom-block decomposition (what each GPU does at N>1): partial passes sum to the whole
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], g.num_atoms)
    cut = g.num_atoms // 3
    ec = int(rowptr[cut])
    fa = torch.zeros_like(f)
    ea = torch.zeros_like(e)
    for (lo, hi, a0, a1) in ((0, ec, 0, cut), (ec, g.num_edges, cut, g.num_atoms)):
        pgp = m.prepare_graph(ei[:, lo:hi], types, g.num_atoms, sv[lo:hi])
        ep, fp = m.energy_forces(pos, pgp)
        fa += fp.cpu().double()
        ea[a0:a1] = ep.cpu().double()[a0:a1]
    assert (fa - f).abs().max().item() < 1e-4 * fscale
    assert (ea - e).abs().max().item() < 1e-5 * max(1.0, e.abs().max().item())
    # oracle on a contiguous sample of center atoms (exact by strict locality)
    a1 = 200
    e1 = int(rowptr[a1])
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ocfg = dict(cfg)
    out = R.allegro_energy_forces(ocfg, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()},
                                  torch.tensor(g.pos), torch.tensor(g.edge_index[:, :e1]), torch.tensor(g.types),
                                  torch.tensor(g.shift_vec()[:e1]))
    assert (out["atomic_energy"].reshape(-1)[:a1] - e[:a1]).abs().max().item() < 5e-5 * max(1.0, e.abs().max().item())
    pgs = m.prepare_graph(ei[:, :e1], types, g.num_atoms, sv[:e1])
    _, fs = m.energy_forces(pos, pgs)
    assert (fs.cpu().double() - out["forces"]).abs().max().item() < 1e-4 * max(1.0, fscale)


def test_rotation_equivariance_fp64(dev):
    """Energies invariant, forces covariant under a random rotation + inversion (small periodic cell rotated
    together with its shift vectors)."""
    fx = load_model_fixture("c2", torch.float64)
    m = model_from_fixture(fx, torch.float64, device=dev)
    data, sv = fixture_data(fx, torch.float64, dev)
    g0 = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    e0, f0 = m.energy_forces(data["pos"], g0)
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(5)))
    q = -q  # include an inversion
    q = q.to(dev)
    g1 = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv @ q.T)
    e1, f1 = m.energy_forces(data["pos"] @ q.T, g1)
    assert (e1 - e0).abs().max().item() < 1e-9
    assert (f1 - f0 @ q.T).abs().max().item() < 1e-9


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 5e-5)])
def test_ragged_graph_two_species_vs_oracle(dtype, tol, dev):
    """Moments + packed two-edge kernels (u = S = 64) on ragged segments: odd degrees and an isolated atom."""
    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    rng = np.random.default_rng(5)
    n = 40
    pos = rng.uniform(0, 10.5, size=(n, 3))
    pos[n - 1] = [40.0, 40.0, 40.0]
    cell = np.eye(3) * 80.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    deg = np.bincount(ei[0], minlength=n)
    assert (deg % 2 == 1).any() and deg[n - 1] == 0
    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=2, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=16, scalar_embed_mlp_hidden_layers_width=32, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=32, avg_num_neighbors=float(deg.mean()), seed=11,
               model_dtype={torch.float64: "float64", torch.float32: "float32"}[dtype])
    m = HipAllegroModel(**cfg).to(dev)
    types = torch.tensor(rng.integers(0, 2, size=n))
    sv = torch.tensor(shift @ cell, dtype=dtype)
    g = m.prepare_graph(torch.tensor(ei).to(dev), types.to(dev), n, sv.to(dev))
    e, f = m.energy_forces(torch.tensor(pos, dtype=dtype, device=dev), g)
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=dtype), torch.tensor(ei), types, sv)
    for got, want in ((e.cpu(), ref["atomic_energy"].reshape(-1)), (f.cpu(), ref["forces"])):
        assert (got - want).abs().max().item() <= tol * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("l_max,dtype,tol", [(1, torch.float32, 5e-5), (2, torch.float32, 5e-5), (3, torch.float32, 5e-5),
                                             (1, torch.float64, 1e-9), (2, torch.float64, 1e-9)])
def test_fast_path_all_lmax_ragged_and_long_segments_vs_oracle(l_max, dtype, tol, dev):
    """The default fast path -- moments TP kernels + fused GEMM chains (all MLP widths 64, u = S = 64, 2 layers) -- at
    every l_max it is specialised for, on a graph mixing sparse ragged segments, an isolated atom and a dense
    cluster whose degrees (69, odd) exceed the 64-edge staging chunk of the reverse kernels."""
    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    rng = np.random.default_rng(21)
    r_max = 6.0
    sparse = rng.uniform(0, 10.5, size=(40, 3)) * (r_max / 3.4)
    grid = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(3), indexing="ij"), -1).reshape(-1, 3)[:70]
    dense = grid * 1.2 + rng.uniform(-0.05, 0.05, size=(70, 3)) + 40.0  # (1.2 A spacing: well-conditioned in fp32)
    pos = np.concatenate([sparse, dense, [[75.0, 75.0, 75.0]]])
    n = len(pos)
    cell = np.eye(3) * 120.0
    ei, shift = G.neighbor_list_pbc(pos, cell, r_max)
    deg = np.bincount(ei[0], minlength=n)
    assert deg.max() == 69 and (deg > 64).sum() > 32 and deg[n - 1] == 0 and (deg[:40] % 2 == 1).any()
    cfg = dict(type_names=["A", "B"], r_max=r_max, l_max=l_max, num_layers=2, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=32, scalar_embed_mlp_hidden_layers_width=64, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=float(deg.mean()), seed=5,
               model_dtype={torch.float64: "float64", torch.float32: "float32"}[dtype])
    m = HipAllegroModel(**cfg).to(dev)
    types = torch.tensor(rng.integers(0, 2, size=n))
    sv = torch.tensor(shift @ cell, dtype=dtype)
    g = m.prepare_graph(torch.tensor(ei).to(dev), types.to(dev), n, sv.to(dev))
    e, f = m.energy_forces(torch.tensor(pos, dtype=dtype, device=dev), g)
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=dtype), torch.tensor(ei), types, sv)
    if dtype == torch.float64:
        for got, want in ((e.cpu(), ref["atomic_energy"].reshape(-1)), (f.cpu(), ref["forces"])):
            assert torch.isfinite(got).all()
            assert (got - want).abs().max().item() <= tol * max(1.0, float(want.abs().max()))
        return
    # fp32: 69-neighbour sums of O(1e3) terms are ill-conditioned, so the yardstick is the fp64 oracle on the SAME
    # (upcast) weights: the HIP path may not be further from it than the fp32 CPU oracle is (x2 + a small floor)
    cfg64 = dict(cfg, model_dtype="float64")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = R.allegro_energy_forces(cfg64, sd64, torch.tensor(pos), torch.tensor(ei), types, sv.double())
    for got, w32, w64 in ((e.cpu(), ref["atomic_energy"].reshape(-1), ref64["atomic_energy"].reshape(-1)),
                          (f.cpu(), ref["forces"], ref64["forces"])):
        assert torch.isfinite(got).all()
        scale = max(1.0, float(w64.abs().max()))
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu32 = (w32.double() - w64).abs().max().item()
        assert err_hip <= 2.0 * err_cpu32 + 1e-5 * scale, (err_hip, err_cpu32, scale)


@pytest.mark.parametrize("l_max,L,u,S,dtype,tol,force,we,slot", [
    (1, 3, 64, 64, torch.float32, 5e-5, False, 64, False), (2, 3, 64, 64, torch.float64, 1e-9, False, 64, True),
    (2, 3, 64, 64, torch.float64, 1e-9, False, 64, False),
    (3, 3, 128, 128, torch.float64, 1e-9, False, 64, False), (3, 3, 128, 128, torch.float64, 1e-9, False, 128, True),
    (2, 2, 128, 64, torch.float32, 5e-5, False, 64, True), (2, 2, 128, 64, torch.float32, 5e-5, False, 64, False),
    (2, 2, 64, 64, torch.float64, 1e-9, True, 64, True), (3, 3, 64, 128, torch.float32, 5e-5, False, 64, False),
    (3, 3, 64, 128, torch.float32, 5e-5, False, 128, True)])
def test_operator_path_vs_oracle(l_max, L, u, S, dtype, tol, force, we, slot, dev, monkeypatch):
    """aa_tp_op.hip (per-atom operator form of the tensor-product track) on hardware: 3-layer stacks, 64- and
    128-channel models (one wave per 64-channel slice), all l_max, both dtypes, ragged segments incl. degrees > 64.
    `slot`: the slot form of the linear layers (needs scalar_embed_mlp width `we` == S) or the unfolded single-layer pipeline
    (`we` != S, or aa_plan_options.no_slot_form where the shapes would allow the slot form)."""
    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    if force:
        monkeypatch.setenv("AA_TP_OP", "1")
    rng = np.random.default_rng(31)
    r_max = 6.0
    sparse = rng.uniform(0, 10.5, size=(30, 3)) * (r_max / 3.4)
    grid = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(3), indexing="ij"), -1).reshape(-1, 3)[:70]
    dense = grid * 1.2 + rng.uniform(-0.05, 0.05, size=(70, 3)) + 40.0
    pos = np.concatenate([sparse, dense, [[75.0, 75.0, 75.0]]])
    n = len(pos)
    cell = np.eye(3) * 120.0
    ei, shift = G.neighbor_list_pbc(pos, cell, r_max)
    deg = np.bincount(ei[0], minlength=n)
    assert deg.max() == 69 and deg[n - 1] == 0
    name = {torch.float64: "float64", torch.float32: "float32"}[dtype]
    cfg = dict(type_names=["A", "B"], r_max=r_max, l_max=l_max, num_layers=L, num_scalar_features=S, num_tensor_features=u,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=128 if S == 128 else 32,  # (128: the 16-lanes-per-edge form of the geometry reverse)
               scalar_embed_mlp_hidden_layers_width=we, allegro_mlp_hidden_layers_width=S,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=float(deg.mean()), seed=5, model_dtype=name)
    if not slot and we == S:
        monkeypatch.setenv("AA_NO_SLOT_FORM", "1")
        monkeypatch.setenv("AA_OP_RECOMPUTE_BVECS", "1")  # (and the layer-0 reverse recomputing the per-atom vectors B_l)
        monkeypatch.setenv("AA_READOUT_TWO_PASS", "1")    # (and d E / d readout hidden by its own kernel)
    if (l_max + L + u // 64 + int(slot)) % 2 == 0:  # half of the cases: the env projections as batched linear-layer launches
        monkeypatch.setenv("AA_OP_PROJ", "1")
    m = HipAllegroModel(**cfg).to(dev)
    d = m.describe_plan()
    assert d["operator_path"] and d["slot_form"] == slot, d
    types = torch.tensor(rng.integers(0, 2, size=n))
    sv = torch.tensor(shift @ cell, dtype=dtype)
    g = m.prepare_graph(torch.tensor(ei).to(dev), types.to(dev), n, sv.to(dev))
    e, f = m.energy_forces(torch.tensor(pos, dtype=dtype, device=dev), g)
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=dtype), torch.tensor(ei), types, sv)
    if dtype == torch.float64:
        for got, want in ((e.cpu(), ref["atomic_energy"].reshape(-1)), (f.cpu(), ref["forces"])):
            assert torch.isfinite(got).all()
            assert (got - want).abs().max().item() <= tol * max(1.0, float(want.abs().max()))
        return
    cfg64 = dict(cfg, model_dtype="float64")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = R.allegro_energy_forces(cfg64, sd64, torch.tensor(pos), torch.tensor(ei), types, sv.double())
    for got, w32, w64 in ((e.cpu(), ref["atomic_energy"].reshape(-1), ref64["atomic_energy"].reshape(-1)),
                          (f.cpu(), ref["forces"], ref64["forces"])):
        assert torch.isfinite(got).all()
        scale = max(1.0, float(w64.abs().max()))
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu32 = (w32.double() - w64).abs().max().item()
        assert err_hip <= 2.0 * err_cpu32 + 1e-5 * scale, (err_hip, err_cpu32, scale)


def test_force_gather_is_deterministic_and_matches_atomics(dev):
    """With the transposed CSR the forces are gathered per atom in a fixed order: bit-identical across runs; the
    atomic fallback (no transposed CSR) agrees to rounding."""
    from allegro_amd.nn import PreparedGraph

    fx = load_model_fixture("c2", torch.float32)
    m, g, _, _ = _run(fx, torch.float32, dev)
    data, sv = fixture_data(fx, torch.float32, dev)
    pos = data["pos"]
    runs = [m.energy_forces(pos, g)[1].clone() for _ in range(3)]
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    g_at = PreparedGraph(data["edge_index"], data["atom_types"], pos.shape[0], sv, transposed=False)
    f_at = m.energy_forces(pos, g_at)[1]
    assert (f_at - runs[0]).abs().max().item() <= 2e-5 * max(1.0, float(runs[0].abs().max()))


def test_hip_graph_replay_matches_direct_launches(dev):
    """aa_model_plan_enable_graph: the captured launch sequence replays bit-identically, follows in-place position
    updates, and is re-captured when an argument (here: the graph) changes."""
    fx = load_model_fixture("c2", torch.float32)
    m, g, _, _ = _run(fx, torch.float32, dev)
    data, sv = fixture_data(fx, torch.float32, dev)
    pos = data["pos"].clone()
    e0, f0 = (t.clone() for t in m.energy_forces(pos, g))
    m.enable_hip_graph(True)
    try:
        for _ in range(3):
            e1, f1 = m.energy_forces(pos, g)
            assert torch.equal(e1, e0) and torch.equal(f1, f0)
        pos.add_(0.01 * torch.randn_like(pos))  # in place: same address, new values
        e2, f2 = (t.clone() for t in m.energy_forces(pos, g))
        g2 = m.prepare_graph(data["edge_index"], data["atom_types"], pos.shape[0], sv)  # new buffers -> re-capture
        e3, f3 = (t.clone() for t in m.energy_forces(pos, g2))
    finally:
        m.enable_hip_graph(False)
    e4, f4 = m.energy_forces(pos, g)
    assert not torch.equal(f2, f0)
    assert torch.equal(e2, e4) and torch.equal(f2, f4) and torch.equal(e3, e4) and torch.equal(f3, f4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_graph_without_edges_gives_shifts_and_zero_forces(dtype, dev):
    from allegro_amd.nn import HipAllegroModel

    cfg = dict(type_names=["A", "B"], r_max=3.4, l_max=2, num_layers=2, num_scalar_features=64, num_tensor_features=64,
               radial_chemical_embed={"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8},
               radial_chemical_embed_dim=64, scalar_embed_mlp_hidden_layers_width=64, allegro_mlp_hidden_layers_width=64,
               readout_mlp_hidden_layers_width=64, avg_num_neighbors=10.0, seed=11,
               model_dtype={torch.float64: "float64", torch.float32: "float32"}[dtype], per_type_energy_shifts=[1.5, -2.0])
    m = HipAllegroModel(**cfg).to(dev)
    pos = torch.tensor([[0.0, 0, 0], [10, 0, 0], [0, 10, 0]], dtype=dtype, device=dev)
    g = m.prepare_graph(torch.zeros((2, 0), dtype=torch.long, device=dev), torch.tensor([0, 1, 0], device=dev), 3, None)
    e, f = m.energy_forces(pos, g)
    assert e.cpu().tolist() == [1.5, -2.0, 1.5] and float(f.abs().max()) == 0.0


@pytest.mark.parametrize("name,dtype,tol", [("t_coupled", torch.float64, 1e-9), ("c5_small", torch.float64, 1e-9),
                                            ("c2", torch.float32, 5e-5), ("c1_L2", torch.float32, 5e-5),
                                            ("c2_spline", torch.float32, 5e-5)])
def test_virial_matches_oracle_strain_derivative(name, dtype, tol, dev, forward_mode):
    """aa_model_virial (strain derivative, stress * volume) vs autograd through the oracle with strained positions and
    shifts; also identical for the gather and the atomic force layouts, and symmetric.  `c2`, `c1_L2` (channel-padded)
    and `c2_spline` take the fused forward in "auto" mode (asserted on the launch list): the virial reads the unit
    vectors that kernel leaves in the workspace."""
    from oracle import restatement as R
    from allegro_amd.nn import PreparedGraph

    fx = load_model_fixture(name, dtype)
    m, g, _, _ = _run(fx, dtype, dev)
    w = m.virial(g).cpu()
    if dtype == torch.float32:
        names = _launches(m, g, fixture_data(fx, dtype, dev)[0]["pos"])
        assert ("fused_fwd" in names) == (forward_mode != "staged"), names
        m.energy_forces(fixture_data(fx, dtype, dev)[0]["pos"], g)
        assert torch.equal(m.virial(g).cpu(), w)  # bit-reproducible (fixed summation order)
    cfg = dict(fx["cfg"])
    cfg["model_dtype"] = "float64"
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in fx["sd"].items()}
    ref = R.allegro_virial(cfg, sd, fx["pos"].double(), fx["edge_index"], fx["types"],
                           None if fx["shift_vec"] is None else fx["shift_vec"].double())
    scale = max(1.0, float(ref.abs().max()))
    assert (w.double() - ref).abs().max().item() <= tol * scale
    assert (w - w.T).abs().max().item() <= 10 * tol * scale
    data, sv = fixture_data(fx, dtype, dev)
    g_at = PreparedGraph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv, transposed=False)
    m.energy_forces(data["pos"], g_at)
    assert (m.virial(g_at).cpu() - w).abs().max().item() <= 10 * tol * scale


@pytest.mark.parametrize("species,embed,dtype,tol", [
    (2, "Bessel", torch.float32, 5e-5), (3, "Bessel", torch.float32, 5e-5), (2, "Spline", torch.float32, 5e-5),
    (3, "Spline", torch.float32, 5e-5), (2, "Spline", torch.float64, 1e-9)])
def test_fast_path_species_and_embedding_variants_vs_oracle(species, embed, dtype, tol, dev):
    """u = S = S0 = 64, 8 radial functions: with <= 2 species the reverse of the two-body embedding is folded into the
    last reverse chain through the per-class table (Bessel: type-embedding x basis weights, spline: class weights);
    with 3 species the stand-alone reverse kernel runs.  Both against the oracle on a ragged periodic graph."""
    from oracle import restatement as R
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel

    rng = np.random.default_rng(31)
    n = 48
    pos = rng.uniform(0, 11.0, size=(n, 3))
    cell = np.eye(3) * 11.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.6)
    deg = np.bincount(ei[0], minlength=n)
    rce = ({"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8} if embed == "Bessel" else
           {"_target_": "allegro.nn.TwoBodySplineScalarEmbed", "num_splines": 8, "spline_span": 5})
    cfg = dict(type_names=["A", "B", "C"][:species], r_max=3.6, l_max=2, num_layers=2, num_scalar_features=64,
               num_tensor_features=64, radial_chemical_embed=rce, scalar_embed_mlp_hidden_layers_width=64,
               allegro_mlp_hidden_layers_width=64, readout_mlp_hidden_layers_width=64,
               per_edge_type_cutoff={"A": 3.6, "B": {"A": 3.2, "B": 3.0}} if species == 2 else None,
               avg_num_neighbors=float(deg.mean()), seed=17,
               model_dtype={torch.float64: "float64", torch.float32: "float32"}[dtype])
    m = HipAllegroModel(**cfg).to(dev)
    types = torch.tensor(rng.integers(0, species, size=n))
    sv = torch.tensor(shift @ cell, dtype=dtype)
    g = m.prepare_graph(torch.tensor(ei).to(dev), types.to(dev), n, sv.to(dev))
    e, f = m.energy_forces(torch.tensor(pos, dtype=dtype, device=dev), g)
    w = m.virial(g).cpu()
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=dtype), torch.tensor(ei), types, sv)
    for got, want in ((e.cpu(), ref["atomic_energy"].reshape(-1)), (f.cpu(), ref["forces"])):
        assert (got - want).abs().max().item() <= tol * max(1.0, float(want.abs().max()))
    cfg64 = dict(cfg, model_dtype="float64")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    wref = R.allegro_virial(cfg64, sd64, torch.tensor(pos), torch.tensor(ei), types, sv.double())
    assert (w.double() - wref).abs().max().item() <= 4 * tol * max(1.0, float(wref.abs().max()))


@pytest.mark.parametrize("over", [dict(num_tensor_features=96), dict(num_tensor_features=16, readout_mlp_hidden_layers_width=32),
                                  dict(num_tensor_features=32, num_layers=3, allegro_mlp_hidden_layers_width=48)])
def test_zero_padded_stacks_match_the_oracle_and_their_narrow_kernels(over, dev, monkeypatch):
    """Channel / hidden-width padding on hardware (aa_model_plan_create): the padded evaluation meets the fp64-oracle
    criterion of the narrow model and agrees with the narrow kernels (AA_NO_PAD=1)."""
    import numpy as np

    from allegro_amd.nn import HipAllegroModel
    from tests.fastpath_utils import _cfg, _ragged
    from tests.fastpath_utils import _vs_oracle64

    pos, cell, ei, shift, types = _ragged(dims=(6, 6, 5), keep=0.93, seed=8)
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    cfg = _cfg("bessel", True, avg=float(deg.mean()))
    cfg.update(over)
    m = _vs_oracle64(cfg, pos, cell, ei, shift, types, None, dev)
    monkeypatch.setenv("AA_NO_PAD", "1")
    m2 = HipAllegroModel(**cfg).to(dev)
    m2.load_state_dict(m.state_dict())
    args = (torch.tensor(ei).to(dev), torch.tensor(types).to(dev), pos.shape[0], torch.tensor(shift @ cell, dtype=torch.float32, device=dev))
    p32 = torch.tensor(pos, dtype=torch.float32, device=dev)
    f1 = m.energy_forces(p32, m.prepare_graph(*args))[1]
    f2 = m2.energy_forces(p32, m2.prepare_graph(*args))[1]
    assert (f1 - f2).abs().max().item() < 2e-5 * max(1.0, float(f2.abs().max()))
