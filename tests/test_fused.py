"""Fused per-atom-tile kernels (allegro_amd/csrc/aa_fused.hip): the whole forward of the standard 2-layer, 64-wide fp32
stack in one launch, one wave per center atom's edge tile.

CPU: the unmodified kernel source in the test-only emulation build, against the reference's golden vectors, against
the fp64 oracle on ragged graphs (partial tiles, an atom without edges, two species, per-type scale/shift), and on
atom-block sub-ranges (the multi-GPU partition).  GPU: the same checks on hardware plus the staged pipeline as A/B.
The fused path needs `aa_graph.max_degree <= 128` (up to 32: one wave per atom; above: teams of waves; larger segments fall
back to the staged pipeline) and is the DEFAULT
forward wherever the graph allows it; AA_FUSED=0 selects the staged pipeline."""
import numpy as np
import pytest
import torch

from allegro_amd import graph as G
from allegro_amd.nn import HipAllegroModel
from tests.golden_utils import load_model_fixture
from tests.hip_utils import emu_lib, fixture_data, model_from_fixture


def _opt_in(monkeypatch, mode="tile32"):
    """The fused forward is the default wherever the graph allows it (AA_FUSED unset); AA_FUSED=1 states it."""
    monkeypatch.setenv("AA_FUSED", "1")


from tests.fastpath_utils import _assert_launched, _cfg, _ragged, _vs_oracle64  # noqa: E402,F401


def _check_vs_oracle64(cfg, pos, cell, ei, shift, types, lib, dev, blocks=None):
    """HIP fp32 (fused) may not be further from the fp64 oracle on the same (upcast) weights than the fp32 CPU oracle
    is (x2 + a small floor) -- the criterion of tests/test_hip_model.py for ill-conditioned fp32 sums."""
    from oracle import restatement as R

    n = pos.shape[0]
    m = HipAllegroModel(**cfg).to(dev)
    if lib is not None:
        m._bind_library(lib)
    sv = torch.tensor(shift @ cell, dtype=torch.float32)
    tt = torch.tensor(types)
    g = m.prepare_graph(torch.tensor(ei).to(dev), tt.to(dev), n, sv.to(dev))
    assert 0 < g.max_degree <= 32
    e, f = m.energy_forces(torch.tensor(pos, dtype=torch.float32, device=dev), g)
    e, f = e.cpu(), f.cpu()
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref32 = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=torch.float32), torch.tensor(ei), tt, sv)
    cfg64 = dict(cfg, model_dtype="float64")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = R.allegro_energy_forces(cfg64, sd64, torch.tensor(pos), torch.tensor(ei), tt, sv.double())
    for got, w32, w64 in ((e, ref32["atomic_energy"].reshape(-1), ref64["atomic_energy"].reshape(-1)),
                          (f, ref32["forces"], ref64["forces"])):
        assert torch.isfinite(got).all()
        scale = max(1.0, float(w64.abs().max()))
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu32 = (w32.double() - w64).abs().max().item()
        assert err_hip <= 2.0 * err_cpu32 + 1e-5 * scale, (err_hip, err_cpu32, scale)
    if blocks:
        # atom-block decomposition (allegro_amd/dist.py): the sum over blocks reproduces the full evaluation
        rowptr = G.csr_from_sorted_centers(ei[0], n)
        cuts = [0] + [int(np.searchsorted(rowptr, rowptr[-1] * k / blocks, side="left")) for k in range(1, blocks)] + [n]
        fa, ea = torch.zeros_like(f), torch.zeros_like(e)
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            lo, hi = int(rowptr[a0]), int(rowptr[a1])
            if hi == lo:
                continue
            gb = m.prepare_graph(torch.tensor(ei[:, lo:hi]).to(dev), tt.to(dev), n, sv[lo:hi].to(dev))
            eb, fb = m.energy_forces(torch.tensor(pos, dtype=torch.float32, device=dev), gb)
            fa += fb.cpu()
            ea[a0:a1] = eb.cpu()[a0:a1]
            # atoms outside the block carry no edges: E_i = shift of their type
            shifts = torch.tensor(cfg.get("per_type_energy_shifts", [0.0, 0.0]), dtype=torch.float32)[tt]
            outside = torch.ones(n, dtype=torch.bool)
            outside[a0:a1] = False
            assert (eb.cpu()[outside] - shifts[outside]).abs().max() < 1e-6
        assert (fa - f).abs().max().item() < 2e-5 * max(1.0, float(f.abs().max()))
        iso = np.bincount(ei[0], minlength=n) == 0
        assert (ea - e)[~torch.tensor(iso)].abs().max().item() < 1e-5 * max(1.0, float(e.abs().max()))
    return m, g


# (the emulated 64-wide model takes ~18 s per case: one fixture here, the GPU tests below run every combination)
@pytest.mark.parametrize("mode,name", [("tile32", "c2_uncoupled")])
def test_fused_forward_matches_reference_golden_emulated(mode, name, monkeypatch):
    _opt_in(monkeypatch, mode)
    fx = load_model_fixture(name, torch.float32)
    m = model_from_fixture(fx, torch.float32, emu_lib())
    data, sv = fixture_data(fx, torch.float32)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    assert 0 < g.max_degree <= 32
    e, f = m.energy_forces(data["pos"], g)
    for got, want in ((e, fx["out"]["atomic_energy"].reshape(-1)), (f, fx["out"]["forces"])):
        assert (got - want).abs().max().item() <= 5e-5 * max(1.0, float(want.abs().max()))


def _wide_vs_narrow(name, lib, dev, monkeypatch):
    """One-species plans take the two-waves-per-SIMD form of the one-tile pass (aa_fused8.hip: its own weight program, merged
    latent-1 / readout phase, w0 re-read from the stored rows) as two four-wave workgroups per CU, or -- AA_FUSED_NARROW=2 -- one
    eight-wave workgroup; AA_FUSED_NARROW=1 keeps the one-wave-per-SIMD kernel.  Same function: all against the reference's golden
    vectors, the two workgroup forms bit-equal, the old kernel to rounding (the summation orders differ)."""
    fx = load_model_fixture(name, torch.float32)
    out = {}
    # "3": two four-wave workgroups per CU (the default from 4 atoms per CU on) | "5": ... env projections on the matrix cores (A/B form)
    # | "2": one eight-wave workgroup, the readout-reverse chain in its tail (the default from 64 atoms per CU on) | "1": the one-wave-per-SIMD kernel
    for narrow in ("3", "5", "2", "1"):
        monkeypatch.setenv("AA_FUSED_NARROW", narrow)
        m = model_from_fixture(fx, torch.float32, lib, device=dev)
        assert m.describe_plan()["fused_wide"] == (narrow != "1")
        data, sv = fixture_data(fx, torch.float32, dev)
        g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
        assert 0 < g.max_degree <= 32
        e, f = m.energy_forces(data["pos"], g)
        assert "fused_fwd" in _launches(m, data, g)
        out[narrow] = (e.cpu().clone(), f.cpu().clone())
        for got, want in ((out[narrow][0], fx["out"]["atomic_energy"].reshape(-1)), (out[narrow][1], fx["out"]["forces"])):
            assert (got - want).abs().max().item() <= 5e-5 * max(1.0, float(want.abs().max()))
    for other in ("1", "5"):
        for a, b in zip(out["3"], out[other]):
            assert (a - b).abs().max().item() <= 2e-5 * max(1.0, float(b.abs().max()))
    # the eight-wave form runs the same forward arithmetic in the same order (energies bit-equal) and, when forces are requested, the
    # readout-reverse chain in its tail instead of as a launch of its own (forces to rounding)
    assert torch.equal(out["3"][0], out["2"][0])
    assert (out["3"][1] - out["2"][1]).abs().max().item() <= 2e-5 * max(1.0, float(out["2"][1].abs().max()))


def test_eight_wave_form_matches_the_four_wave_form_and_the_golden_vectors_emulated(monkeypatch):
    _opt_in(monkeypatch)
    _wide_vs_narrow("c2", emu_lib(), torch.device("cpu"), monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c2_spline", "c2_l1", "c2_uncoupled", "c1_L2"])
def test_eight_wave_form_matches_the_four_wave_form_and_the_golden_vectors_on_gpu(name, monkeypatch):
    _opt_in(monkeypatch)
    _wide_vs_narrow(name, None, torch.device("cuda"), monkeypatch)


@pytest.mark.parametrize("mode,embed,coupling", [("tile32", "spline", False)])
def test_fused_forward_ragged_graph_vs_fp64_oracle_emulated(mode, embed, coupling, monkeypatch):
    _opt_in(monkeypatch, mode)
    pos, cell, ei, shift, types = _ragged()
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    assert 20 <= deg.max() <= 32 and deg.min() == 0 and len(set(deg.tolist())) > 6, deg
    _check_vs_oracle64(_cfg(embed, coupling, avg=float(deg.mean())), pos, cell, ei, shift, types, emu_lib(), torch.device("cpu"),
                       blocks=3)


def _three_species(dims, seed, embed, lib, dev, monkeypatch):
    monkeypatch.delenv("AA_FUSED", raising=False)  # automatic: the 32-edge-tile fused forward
    pos, cell, ei, shift, _ = _ragged(dims=dims, keep=0.9, seed=seed)
    types = np.random.default_rng(seed).integers(0, 3, size=pos.shape[0])
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    cfg = _cfg(embed, True, avg=float(deg.mean()), scale_shift=False)
    cfg.update(type_names=["A", "B", "C"], per_type_energy_scales=[1.3, 0.6, 0.9], per_type_energy_shifts=[-2.0, 0.25, 1.0])
    m, g = _check_vs_oracle64(cfg, pos, cell, ei, shift, types, lib, dev)
    assert "fused_fwd" in _launches(m, {"pos": torch.tensor(pos, dtype=torch.float32, device=dev)}, g)


def test_three_species_run_the_fused_forward_emulated(monkeypatch):
    """Three species (the 3 x 3 two-body table is 18 KB of the kernel's LDS; more fall back to the staged forward)."""
    _three_species((3, 3, 2), 4, "bessel", emu_lib(), torch.device("cpu"), monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("embed", ["bessel", "spline"])
def test_three_species_run_the_fused_forward_on_gpu(embed, monkeypatch):
    _three_species((8, 8, 7), 6, embed, None, torch.device("cuda:0"), monkeypatch)


def _mixed_degree_cluster(cut=2.2, per_class=(3, 3, 2, 1)):
    """110 atoms on a jittered 0.7 grid + one isolated atom; the edge segments of a few center atoms of every tile class
    (1..32, 33..64, 65..96, 97..128 edges) are kept, the other atoms are neighbors only (emulation time)."""
    rng = np.random.default_rng(3)
    grid = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3)[:110]
    pos = grid * 0.7 + rng.uniform(-0.05, 0.05, size=(110, 3)) + 20.0
    pos = np.concatenate([pos, [[50.0, 50.0, 50.0]]])
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, cut)
    deg = np.bincount(ei[0], minlength=111)
    pick = []
    for (lo, hi), k in zip(((1, 32), (33, 64), (65, 96), (97, 128)), per_class):
        pick += [int(i) for i in np.argsort(deg, kind="stable") if lo <= deg[i] <= hi][:k]
    keep = np.isin(ei[0], pick)
    return pos, cell, ei[:, keep], shift[keep], rng.integers(0, 2, size=111), sorted(int(deg[i]) for i in pick)


def _teams_case(lib, dev, monkeypatch, cut=2.2, per_class=(3, 3, 2, 1), form=None):
    """Segments of more than one 32-edge tile run the TEAM form of the fused forward (2 / 4 waves per atom, per-atom sums
    completed through LDS): against the fp64 oracle, against the staged pipeline on the same graph, bit-reproducible although
    the slot assignment comes from atomic counters, and the launch list names the kernel.  `form`: None = what the plan selects
    (since round 5 the MIXED form: the one-tile kernel over all atoms skipping the long ones + the team kernel over the long ones
    only), "2" = the team kernel for every atom, "4" = the mixed form forced."""
    if form is None:
        monkeypatch.delenv("AA_FUSED", raising=False)
    else:
        monkeypatch.setenv("AA_FUSED", form)
    pos, cell, ei, shift, types, degs = _mixed_degree_cluster(cut, per_class)
    assert degs[0] <= 32 and any(32 < d <= 64 for d in degs) and any(64 < d <= 128 for d in degs), degs
    cfg = _cfg(avg=float(np.mean(degs)), scale_shift=True)
    m = _vs_oracle64(cfg, pos, cell, ei, shift, types, lib, dev)
    sv = torch.tensor(shift @ cell, dtype=torch.float32, device=dev)
    g = m.prepare_graph(torch.tensor(ei).to(dev), torch.tensor(types).to(dev), pos.shape[0], sv)
    assert 32 < g.max_degree <= 128
    p = torch.tensor(pos, dtype=torch.float32, device=dev)
    e1, f1 = (t.clone() for t in m.energy_forces(p, g))
    v1 = m.virial(g).clone()
    e2, f2 = (t.clone() for t in m.energy_forces(p, g))
    assert torch.equal(e1, e2) and torch.equal(f1, f2) and torch.equal(v1, m.virial(g))
    import bench

    names = [s[0] for s in bench.profile_stages(m, p, g, reps=1)]
    assert "fused_fwd" in names, names
    monkeypatch.setenv("AA_FUSED", "0")
    ms = HipAllegroModel(**cfg).to(dev)
    ms.load_state_dict(m.state_dict())
    if lib is not None:
        ms._bind_library(lib)
    es, fs = ms.energy_forces(p, g)
    vs = ms.virial(g).clone()
    names = [s[0] for s in bench.profile_stages(ms, p, g, reps=1)]
    assert "fused_fwd" not in names, names
    assert (v1 - vs).abs().max().item() <= 5e-5 * max(1.0, float(vs.abs().max()))  # strain derivative of the same step
    assert (e1 - es).abs().max().item() <= 2e-5 * max(1.0, float(es.abs().max()))
    assert (f1 - fs).abs().max().item() <= 2e-5 * max(1.0, float(fs.abs().max()))


def test_segments_of_several_tiles_run_the_team_form_emulated(monkeypatch):
    # (what the plan selects: the mixed form; the team kernel for every atom is compared bit by bit in the next test and runs
    #  through this case on the GPU)
    _teams_case(emu_lib(), torch.device("cpu"), monkeypatch, form=None)


def test_mixed_and_team_forms_agree_bitwise_emulated(monkeypatch):
    """The mixed form (one-tile kernel, long atoms skipped + team kernel over the long atoms only) and the team form for every atom
    run the same per-tile arithmetic in the same order: energies and forces are bit-equal; a graph with a single long atom among
    short ones (the MD case: thermal disorder) is selected for the mixed form by its fill."""
    pos, cell, ei, shift, types, degs = _mixed_degree_cluster(2.2, (6, 1, 0, 0))
    assert sum(d > 32 for d in degs) == 1
    cfg = _cfg(avg=float(np.mean(degs)), scale_shift=True)
    outs = []
    for form in (None, "2", "0"):
        if form is None:
            monkeypatch.delenv("AA_FUSED", raising=False)
        else:
            monkeypatch.setenv("AA_FUSED", form)
        m = HipAllegroModel(**cfg)
        m._bind_library(emu_lib())
        if outs:
            m.load_state_dict(outs[0][2])
        g = m.prepare_graph(torch.tensor(ei), torch.tensor(types), pos.shape[0], torch.tensor(shift @ cell, dtype=torch.float32))
        p = torch.tensor(pos, dtype=torch.float32)
        e, f = (t.clone() for t in m.energy_forces(p, g))
        m.check()
        import bench

        names = [s[0] for s in bench.profile_stages(m, p, g, reps=1)]
        assert ("fused_fwd" in names) == (form != "0"), (form, names)
        outs.append((e, f, m.state_dict()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][0] - outs[2][0]).abs().max().item() <= 2e-5 * max(1.0, float(outs[2][0].abs().max()))
    assert (outs[0][1] - outs[2][1]).abs().max().item() <= 2e-5 * max(1.0, float(outs[2][1].abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("cut,per_class", [(2.2, (3, 3, 2, 1)), (2.6, (0, 40, 70, 10)), (2.0, (40, 70, 10, 0))])
def test_segments_of_several_tiles_run_the_team_form_on_gpu(cut, per_class, monkeypatch):
    _teams_case_gpu(cut, per_class, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("form", [None, "2"])
def test_team_form_against_the_staged_pipeline_on_gpu(form, monkeypatch):
    _teams_case(None, torch.device("cuda:0"), monkeypatch, form=form)


def _teams_case_gpu(cut, per_class, monkeypatch):
    # (the class mix asserted by _teams_case needs all three; the wide selections keep every center atom of the cluster)
    monkeypatch.delenv("AA_FUSED", raising=False)
    dev = torch.device("cuda:0")
    pos, cell, ei, shift, types, degs = _mixed_degree_cluster(cut, per_class)
    cfg = _cfg(avg=float(np.mean(degs)), scale_shift=True)
    m = _vs_oracle64(cfg, pos, cell, ei, shift, types, None, dev)
    sv = torch.tensor(shift @ cell, dtype=torch.float32, device=dev)
    g = m.prepare_graph(torch.tensor(ei).to(dev), torch.tensor(types).to(dev), pos.shape[0], sv)
    assert 32 < g.max_degree <= 128
    p = torch.tensor(pos, dtype=torch.float32, device=dev)
    ref = [t.clone() for t in m.energy_forces(p, g)]
    for _ in range(5):  # slot assignment comes from atomic counters: results must not depend on it
        e, f = m.energy_forces(p, g)
        assert torch.equal(e, ref[0]) and torch.equal(f, ref[1])
    import bench

    assert "fused_fwd" in [s[0] for s in bench.profile_stages(m, p, g, reps=1)]


@pytest.mark.gpu
def test_team_form_is_selected_where_it_pays_on_gpu(monkeypatch):
    """Dense Si boxes (r_max 6: 44 edges per atom, two tiles 69 % full): the team form runs up to 4096 tiles (1 728 atoms),
    the staged forward beyond (4 096 atoms) -- aa_model.hip: use_fused_fwd; both agree with each other."""
    import bench

    monkeypatch.delenv("AA_FUSED", raising=False)
    dev = torch.device("cuda:0")
    for cells, fused in ((6, True), (8, False)):
        g = G.make_si_graph(cells, r_cut=6.0)
        cfg = _cfg(avg=g.num_edges / g.num_atoms, scale_shift=False)
        cfg["r_max"] = 6.0
        m = HipAllegroModel(**cfg).to(dev)
        pg = m.prepare_graph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), g.num_atoms,
                             torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev))
        assert 32 < pg.max_degree <= 64
        pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
        names = [s[0] for s in bench.profile_stages(m, pos, pg, reps=1)]
        assert ("fused_fwd" in names) == fused, (cells, names)
        e, f = (t.clone() for t in m.energy_forces(pos, pg))
        monkeypatch.setenv("AA_FUSED", "0")
        ms = HipAllegroModel(**cfg).to(dev)
        ms.load_state_dict(m.state_dict())
        es, fs = ms.energy_forces(pos, pg)
        monkeypatch.delenv("AA_FUSED", raising=False)
        assert (f - fs).abs().max().item() <= 2e-5 * max(1.0, float(fs.abs().max()))
        assert (e - es).abs().max().item() <= 2e-5 * max(1.0, float(es.abs().max()))


def test_degree_above_128_falls_back_to_the_staged_pipeline_emulated(monkeypatch):
    _opt_in(monkeypatch)
    rng = np.random.default_rng(5)
    grid = np.stack(np.meshgrid(np.arange(6), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3)
    pos = grid * 0.7 + rng.uniform(-0.05, 0.05, size=(150, 3)) + 20.0
    cell = np.eye(3) * 60.0
    ei_all, shift_all = G.neighbor_list_pbc(pos, cell, 6.0)
    deg = np.bincount(ei_all[0], minlength=150)
    assert deg.max() > 128
    keep = np.isin(ei_all[0], [0, 75])  # two center atoms keep their 149-edge segments (emulation time)
    types = rng.integers(0, 2, size=150)
    cfg = _cfg(avg=75.0, scale_shift=False)
    m = _vs_oracle64(cfg, pos, cell, ei_all[:, keep], shift_all[keep], types, emu_lib(), torch.device("cpu"))
    _assert_launched(m, pos, cell, ei_all[:, keep], shift_all[keep], types, present=[], absent=["fused_fwd"])


def _launches(m, data, g):
    import bench

    return [s[0] for s in bench.profile_stages(m, data["pos"], g, reps=1)]


def test_automatic_selection_emulated(monkeypatch):
    """AA_FUSED unset = aa_plan_options.fused_forward 0: the 32-edge-tile fused forward whenever the graph allows it
    (max_degree <= 32), the staged pipeline otherwise and under AA_FUSED=0; an atom block (what a rank of a partition
    owns) too.  All match the reference's golden vectors."""
    monkeypatch.delenv("AA_FUSED", raising=False)
    fx = load_model_fixture("c2_spline", torch.float32)
    m = model_from_fixture(fx, torch.float32, emu_lib())
    data, sv = fixture_data(fx, torch.float32)
    n = data["pos"].shape[0]
    ei = data["edge_index"]
    keep = ei[0] < 12
    gb = m.prepare_graph(ei[:, keep], data["atom_types"], n, None if sv is None else sv[keep])
    assert "fused_fwd" in _launches(m, data, gb)
    eb, fb = m.energy_forces(data["pos"], gb)
    want = fx["out"]["atomic_energy"].reshape(-1)
    assert (eb[:12] - want[:12]).abs().max().item() <= 5e-5 * max(1.0, float(want.abs().max()))
    monkeypatch.setenv("AA_FUSED", "0")
    m0 = model_from_fixture(fx, torch.float32, emu_lib())
    assert "fused_fwd" not in _launches(m0, data, gb)
    e0, f0 = m0.energy_forces(data["pos"], gb)
    assert (e0[:12] - eb[:12]).abs().max().item() < 5e-6 and (f0 - fb).abs().max().item() < 2e-5


@pytest.mark.gpu
def test_automatic_selection_on_gpu(monkeypatch):
    """On hardware: the fused forward runs by default at any size (64 atoms and a 1728-atom box), AA_FUSED=0 selects the
    staged pipeline; results agree with the golden vectors either way."""
    from allegro_amd import graph as G

    dev = torch.device("cuda:0")
    fx = load_model_fixture("c2", torch.float32)
    data, sv = fixture_data(fx, torch.float32, dev)
    for mode in ("auto", "0"):
        if mode == "auto":
            monkeypatch.delenv("AA_FUSED", raising=False)
        else:
            monkeypatch.setenv("AA_FUSED", "0")
        m = model_from_fixture(fx, torch.float32, device=dev)
        g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
        assert ("fused_fwd" in _launches(m, data, g)) == (mode == "auto")
        e, f = m.energy_forces(data["pos"], g)
        for got, want in ((e.cpu(), fx["out"]["atomic_energy"].reshape(-1)), (f.cpu(), fx["out"]["forces"])):
            assert (got - want).abs().max().item() <= 5e-5 * max(1.0, float(want.abs().max()))
    monkeypatch.delenv("AA_FUSED", raising=False)
    big = G.make_si_graph(6)
    m = model_from_fixture(fx, torch.float32, device=dev)
    gb = m.prepare_graph(torch.tensor(big.edge_index).to(dev), torch.zeros(big.num_atoms, dtype=torch.int64, device=dev), big.num_atoms,
                         torch.tensor(big.shift_vec(), dtype=torch.float32, device=dev))
    assert "fused_fwd" in _launches(m, {"pos": torch.tensor(big.pos, dtype=torch.float32, device=dev)}, gb)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tile32"])
@pytest.mark.parametrize("embed,coupling,l_max", [("bessel", True, 2), ("spline", False, 2), ("bessel", True, 1)])
def test_fused_forward_ragged_graph_vs_fp64_oracle_on_gpu(mode, embed, coupling, l_max, monkeypatch):
    _opt_in(monkeypatch, mode)
    pos, cell, ei, shift, types = _ragged(dims=(9, 9, 8), keep=0.93, seed=8)
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    assert 24 <= deg.max() <= 32 and deg.min() == 0, (deg.max(), deg.min())
    _check_vs_oracle64(_cfg(embed, coupling, l_max=l_max, avg=float(deg.mean())), pos, cell, ei, shift, types, None,
                       torch.device("cuda:0"), blocks=4)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tile32"])
@pytest.mark.parametrize("name", ["c2", "c2_spline", "c2_l1", "c2_uncoupled"])
def test_fused_and_staged_forward_agree_on_gpu(mode, name, monkeypatch):
    """A/B on hardware: the same model and graph through the fused kernel (AA_FUSED at plan creation) and through
    the staged pipeline, and both against the reference's golden vectors."""
    _opt_in(monkeypatch, mode)
    dev = torch.device("cuda:0")
    fx = load_model_fixture(name, torch.float32)
    data, sv = fixture_data(fx, torch.float32, dev)
    m = model_from_fixture(fx, torch.float32, device=dev)
    g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
    e, f = m.energy_forces(data["pos"], g)
    monkeypatch.setenv("AA_FUSED", "0")  # the staged pipeline: its chains in the folded form (default) ...
    m2 = model_from_fixture(fx, torch.float32, device=dev)
    e2, f2 = m2.energy_forces(data["pos"], g)
    monkeypatch.setenv("AA_STAGED_NOFOLD", "1")  # ... and with the reference's own layers
    m3 = model_from_fixture(fx, torch.float32, device=dev)
    e3, f3 = m3.energy_forces(data["pos"], g)
    names2, names3 = _launches(m2, data, g), _launches(m3, data, g)
    assert "fused_fwd" not in names2 and "fused_fwd" not in names3 and names2 != names3, (names2, names3)
    for ex, fx_ in ((e2, f2), (e3, f3)):
        assert (e - ex).abs().max().item() < 5e-6 and (f - fx_).abs().max().item() < 2e-5
    for got, want in ((e.cpu(), fx["out"]["atomic_energy"].reshape(-1)), (f.cpu(), fx["out"]["forces"]), (e2.cpu(), fx["out"]["atomic_energy"].reshape(-1)),
                      (f2.cpu(), fx["out"]["forces"]), (f3.cpu(), fx["out"]["forces"])):
        assert (got - want).abs().max().item() <= 5e-5 * max(1.0, float(want.abs().max()))


def _stale_hint_case(lib, dev):
    """A C host that passes a stale aa_graph.max_degree (the hint the fused forward trusts for its tile shape) must get an
    error, not energies that silently ignore edges: every edge of the c2 cell listed twice = 56 edges
    per atom, hint left at 28.  The reference takes any segment length (_strided/_contract.py:195-205)."""
    fx = load_model_fixture("c2", torch.float32)
    m = model_from_fixture(fx, torch.float32, lib, device=dev)
    data, sv = fixture_data(fx, torch.float32, dev)
    ei = torch.cat([data["edge_index"], data["edge_index"]], 1)
    sv2 = torch.cat([sv, sv], 0)
    g = m.prepare_graph(ei, data["atom_types"], data["pos"].shape[0], sv2)
    assert g.max_degree == 56
    e_ok, f_ok = m.energy_forces(data["pos"], g)  # true hint: team form of the fused forward
    m.check()
    assert torch.isfinite(e_ok).all() and torch.isfinite(f_ok).all()
    g.max_degree = 28  # stale
    e_bad, f_bad = m.energy_forces(data["pos"], g)
    with pytest.raises(RuntimeError, match="max_degree"):
        m.check()
    assert torch.isnan(e_bad).all()  # every atom has 56 > 32 edges: none is evaluated on a truncated segment
    # ... and the forces of the SAME step are NaN as well (the reverse pass would otherwise combine workspace rows
    # left over from the valid step above into finite, wrong forces that a host integrates before it sees the status)
    assert torch.isnan(f_bad).all()
    # the condition is reported once; without aa_model_check the NEXT step reports it
    e_bad, _ = m.energy_forces(data["pos"], g)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="56 edges"):
        m.energy_forces(data["pos"], g)
    g.max_degree = 56
    e2, f2 = m.energy_forces(data["pos"], g)
    m.check()
    assert torch.equal(e2, e_ok) and torch.equal(f2, f_ok)
    # the atom-block hint is verified the same way: a block that leaves centers with edges outside
    g.atom_begin, g.atom_end = 8, 40
    e_blk, f_blk = m.energy_forces(data["pos"], g)
    with pytest.raises(RuntimeError, match="atom_begin"):
        m.check()
    assert torch.isnan(e_blk).all() and torch.isnan(f_blk).all()  # outputs of that step are poisoned, not partial


def test_stale_max_degree_hint_fails_loudly_emulated():
    _stale_hint_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_stale_max_degree_hint_fails_loudly():
    _stale_hint_case(None, torch.device("cuda:0"))


def _three_species_long_segments(lib, dev, monkeypatch):
    """Three species AND segments of more than one tile: the 18-KB two-body table plus the team exchange area do not fit
    the 160 KB of LDS, so such graphs must take the staged pipeline -- found in round 4: the plan used to select the team form
    and the launch then failed with "LDS budget exceeded" instead of falling back."""
    monkeypatch.delenv("AA_FUSED", raising=False)
    pos, cell, ei, shift, _, degs = _mixed_degree_cluster(2.2, (2, 2, 1, 0))
    types = np.random.default_rng(8).integers(0, 3, size=pos.shape[0])
    cfg = _cfg(avg=float(np.mean(degs)), scale_shift=False)
    cfg.update(type_names=["A", "B", "C"], per_type_energy_scales=[1.3, 0.6, 0.9], per_type_energy_shifts=[-2.0, 0.25, 1.0])
    m = _vs_oracle64(cfg, pos, cell, ei, shift, types, lib, dev)
    sv = torch.tensor(shift @ cell, dtype=torch.float32, device=dev)
    g = m.prepare_graph(torch.tensor(ei).to(dev), torch.tensor(types).to(dev), pos.shape[0], sv)
    assert 32 < g.max_degree <= 128
    import bench

    names = [s[0] for s in bench.profile_stages(m, torch.tensor(pos, dtype=torch.float32, device=dev), g, reps=1)]
    assert "fused_fwd" not in names, names


def test_three_species_with_long_segments_fall_back_to_the_staged_forward_emulated(monkeypatch):
    _three_species_long_segments(emu_lib(), torch.device("cpu"), monkeypatch)


@pytest.mark.gpu
def test_three_species_with_long_segments_fall_back_to_the_staged_forward_on_gpu(monkeypatch):
    _three_species_long_segments(None, torch.device("cuda:0"), monkeypatch)


@pytest.mark.parametrize("nofold", [False, True])
def test_staged_pipeline_in_folded_and_unfolded_form_vs_fp64_oracle_emulated(nofold, monkeypatch):
    """The staged fp32 pipeline (what graphs with long segments run) evaluates its forward chains in the folded form of the fused
    forward -- a_e / a_0 stored where the embedding / lat_0 used to be (ChainLayer::kept_out), consumers on the folded matrices, the
    last reverse chain one 256 -> 64 layer -- or, under aa_plan_options.staged_no_fold, with the reference's own layers."""
    monkeypatch.setenv("AA_FUSED", "0")
    if nofold:
        monkeypatch.setenv("AA_STAGED_NOFOLD", "1")
    pos, cell, ei, shift, types = _ragged(dims=(3, 3, 2))
    m = _vs_oracle64(_cfg(), pos, cell, ei, shift, types, emu_lib(), torch.device("cpu"))
    _assert_launched(m, pos, cell, ei, shift, types, present=["gc_64x64_64x64_64x256" if nofold else "gc_64x64_64x256",
                                                                "gc_128x64_64x64" if nofold else "gc_128x64"], absent=["fused_fwd"])


def _resident_chain_case(lib, dev, monkeypatch, exact):
    """The latent-0 reverse chain with LDS-resident weights (aa_chain_res.hip) is the general chain kernel's layer evaluated by a
    persistent workgroup per CU: same arithmetic in the same order -- energies and forces bit-equal to AA_CHAIN_STAGED=1."""
    fx = load_model_fixture("c2", torch.float32)
    out = {}
    for staged in ("0", "1"):
        monkeypatch.setenv("AA_CHAIN_STAGED", staged)
        m = model_from_fixture(fx, torch.float32, lib, device=dev)
        data, sv = fixture_data(fx, torch.float32, dev)
        g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
        e, f = m.energy_forces(data["pos"], g)
        out[staged] = (e.cpu().clone(), f.cpu().clone())
        assert (out[staged][1] - fx["out"]["forces"]).abs().max().item() <= 5e-5 * max(1.0, float(fx["out"]["forces"].abs().max()))
    assert torch.equal(out["0"][0], out["1"][0])
    if exact:
        assert torch.equal(out["0"][1], out["1"][1])
    else:  # (on the device the two kernels' operand transforms are contracted into FMAs differently by the compiler: last-bit differences)
        assert (out["0"][1] - out["1"][1]).abs().max().item() <= 2e-6 * max(1.0, float(out["1"][1].abs().max()))


def test_resident_weight_chain_is_bit_equal_to_the_staged_chain_emulated(monkeypatch):
    _opt_in(monkeypatch)
    _resident_chain_case(emu_lib(), torch.device("cpu"), monkeypatch, exact=True)


@pytest.mark.gpu
def test_resident_weight_chain_matches_the_staged_chain_on_gpu(monkeypatch):
    _opt_in(monkeypatch)
    _resident_chain_case(None, torch.device("cuda"), monkeypatch, exact=False)
