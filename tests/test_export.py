"""Export path: the whole step as one C++-registered dispatcher op behind the pair_allegro tensor
contract `[pos, edge_index, atom_types] -> (atomic_energy, total_energy, forces, virial)` (allegro/_compile.py:10-14).
CPU: the extension builds and loads, `torch.export` captures the op through its Meta kernel, and there is no CPU
kernel to fall back to.  GPU: the op reproduces `HipAllegroModel.energy_forces` (also for an unsorted edge list and
from a re-loaded exported program)."""
import io

import pytest
import torch

from tests.golden_utils import load_model_fixture
from tests.hip_utils import emu_lib, fixture_data, model_from_fixture


def _exportable(name, dtype, device, lib=None):
    from allegro_amd.export import ExportableAllegro

    fx = load_model_fixture(name, dtype)
    m = model_from_fixture(fx, dtype, lib, device=device)
    data, sv = fixture_data(fx, dtype, device)
    return fx, m, data, sv, ExportableAllegro(m, device)


def test_native_op_is_registered_and_exportable():
    fx, m, data, sv, ex = _exportable("t_coupled", torch.float64, "cpu", emu_lib())
    schema = str(torch.ops.allegro_amd_native.energy_forces.default._schema)
    assert "Tensor? shift_vec" in schema and "int[] config" in schema and "-> (Tensor, Tensor, Tensor)" in schema
    ep = torch.export.export(ex, (data["pos"], data["edge_index"], data["atom_types"], sv))
    targets = [str(n.target) for n in ep.graph.nodes if n.op == "call_function"]
    assert any("allegro_amd_native.energy_forces" in t for t in targets), targets
    outs = [n for n in ep.graph.nodes if n.op == "output"][0].args[0]
    shapes = [tuple(o.meta["val"].shape) for o in outs]
    N = data["pos"].shape[0]
    assert shapes == [(N, 1), (1, 1), (N, 3), (1, 3, 3)]  # LMP_OUTPUTS: per-atom energy, total energy, forces, virial
    # the program round-trips through the serialized form with its constants (config words, weight blob)
    buf = io.BytesIO()
    torch.export.save(ep, buf)
    buf.seek(0)
    ep2 = torch.export.load(buf)
    assert any("allegro_amd_native.energy_forces" in str(n.target) for n in ep2.graph.nodes if n.op == "call_function")


def test_native_op_has_no_cpu_kernel():
    fx, m, data, sv, ex = _exportable("t_coupled", torch.float64, "cpu", emu_lib())
    with pytest.raises((NotImplementedError, RuntimeError)):
        ex(data["pos"], data["edge_index"], data["atom_types"], sv)


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype,tol", [("c2", torch.float32, 5e-5), ("c1_L2", torch.float32, 5e-5), ("t_peredge", torch.float64, 1e-9),
                                            ("c2_spline", torch.float32, 5e-5)])
def test_native_op_matches_reference_golden_on_gpu(name, dtype, tol, forward_mode):
    dev = torch.device("cuda:0")
    fx, m, data, sv, ex = _exportable(name, dtype, dev)
    ref = fx["out"]
    perm = torch.randperm(data["edge_index"].shape[1], generator=torch.Generator().manual_seed(1)).to(dev)
    # the op's virial directly against the ORACLE's strain derivative (autograd through strained positions + shifts)
    from oracle import restatement as R

    cfg64 = dict(fx["cfg"], model_dtype="float64")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in fx["sd"].items()}
    wref = R.allegro_virial(cfg64, sd64, fx["pos"].double(), fx["edge_index"], fx["types"],
                            None if fx["shift_vec"] is None else fx["shift_vec"].double())
    for ei, s in ((data["edge_index"], sv), (data["edge_index"][:, perm], None if sv is None else sv[perm])):
        e_atom, e_tot, f, vir = ex(data["pos"], ei, data["atom_types"], s)
        # virial = -dE/d(strain), checked against the model's own strain derivative (itself pinned to the oracle's
        # autograd-through-strain in test_hip_model.py / test_emu_kernels.py)
        g = m.prepare_graph(ei, data["atom_types"], data["pos"].shape[0], s)
        m.energy_forces(data["pos"], g)
        w = m.virial(g)
        assert vir.shape == (1, 3, 3)
        do = (vir[0].double().cpu() + wref).abs().max().item()
        assert do <= tol * max(1.0, float(wref.abs().max())), f"virial: native op {vir[0].tolist()} vs oracle {(-wref).tolist()} (max diff {do:.3e})"
        dv = (vir[0] + w).abs().max().item()
        assert dv <= tol * max(1.0, float(w.abs().max())), f"virial: native op {vir[0].tolist()} vs model {(-w).tolist()} (max diff {dv:.3e})"
        for what, got, want in (("energies", e_atom.cpu().reshape(-1), ref["atomic_energy"].reshape(-1)), ("forces", f.cpu(), ref["forces"])):
            err = (got - want).abs().max().item()
            assert err <= tol * max(1.0, float(want.abs().max())), f"{what}: max diff {err:.3e}, finite: {bool(torch.isfinite(got).all())}"
        assert abs(float(e_tot) - float(ref["atomic_energy"].sum())) <= 10 * tol * max(1.0, abs(float(ref["atomic_energy"].sum())))
    # exported program, saved and re-loaded, run on the GPU
    ep = torch.export.export(ex, (data["pos"], data["edge_index"], data["atom_types"], sv))
    buf = io.BytesIO()
    torch.export.save(ep, buf)
    buf.seek(0)
    e2, _, f2, _v2 = torch.export.load(buf).module()(data["pos"], data["edge_index"], data["atom_types"], sv)
    assert (f2.cpu() - ref["forces"]).abs().max().item() <= tol * max(1.0, float(ref["forces"].abs().max()))


@pytest.mark.gpu
def test_native_op_on_a_dense_neighbour_list_matches_the_model_on_gpu():
    """The C++ op sizes its own workspace and decides the forward from `max_degree` like the Python host: on a list with
    33..128 edges per atom (team form of the fused forward) it reproduces `HipAllegroModel.energy_forces` and the virial,
    also with the edges shuffled (the op sorts them)."""
    import numpy as np

    from allegro_amd import graph as G
    from allegro_amd.export import ExportableAllegro
    from allegro_amd.nn import HipAllegroModel
    from tests.fastpath_utils import _cfg

    dev = torch.device("cuda:0")
    g = G.make_si_graph(3, r_cut=6.0)  # 216 atoms, 44.5 edges per atom
    cfg = _cfg(avg=g.num_edges / g.num_atoms, scale_shift=True)
    cfg["r_max"] = 6.0
    m = HipAllegroModel(**cfg).to(dev)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    ei = torch.tensor(g.edge_index, device=dev)
    types = torch.tensor(g.types, device=dev)
    sv = torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev)
    pg = m.prepare_graph(ei, types, g.num_atoms, sv)
    assert 32 < pg.max_degree <= 128
    e, f = m.energy_forces(pos, pg)
    w = m.virial(pg)
    ex = ExportableAllegro(m, dev)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(2)).to(dev)
    for idx in (None, perm):
        ei_x, sv_x = (ei, sv) if idx is None else (ei[:, idx], sv[idx])
        ea, et, fo, vir = ex(pos, ei_x, types, sv_x)
        # (fp32 rounding level: the op's plan and the Python host's are two plan objects with their own weight packing)
        assert (ea.reshape(-1) - e).abs().max().item() <= 5e-6 * max(1.0, float(e.abs().max()))
        assert (fo - f).abs().max().item() <= 5e-6 * max(1.0, float(f.abs().max()))
        assert (vir.reshape(3, 3) + w).abs().max().item() <= 2e-5 * max(1.0, float(w.abs().max()))  # (virial = -dE/d strain)
