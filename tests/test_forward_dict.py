"""CPU (kernels in the test-only emulation build): the AtomicDataDict interface of `HipAllegroModel.forward`.

Regression tests for the implicit graph cache: the cached structure is keyed on the edge list and the
atom types only, everything that depends on the cell is recomputed per call; and for batched periodic input (one cell
per frame, `cell[batch[center]]` per edge as nequip's `with_edge_vectors_` does, with per-frame stress / virial)."""
import numpy as np
import torch

from allegro_amd import graph as G
from tests.golden_utils import load_model_fixture
from tests.hip_utils import emu_lib, model_from_fixture


def _molecule(n=24, box=9.0, r_cut=4.0, seed=3):
    """The geometry of the t_* golden fixtures (oracle/make_golden.py: molecule_graph)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, box, size=(n, 3))
    cell = np.eye(3) * box
    ei, shift = G.neighbor_list_pbc(pos, cell, r_cut)
    types = rng.integers(0, 3, size=n).astype(np.int64)
    return pos, cell, ei, shift, types


def _data(pos, cell, ei, shift, types):
    return {"pos": torch.tensor(pos), "edge_index": torch.tensor(ei), "atom_types": torch.tensor(types),
            "cell": torch.tensor(cell), "edge_cell_shift": torch.tensor(shift, dtype=torch.float64)}


def test_dict_forward_reproduces_the_golden_fixture():
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    out = m(_data(*_molecule()))
    assert (out["forces"] - fx["out"]["forces"]).abs().max() < 1e-9
    assert (out["atomic_energy"] - fx["out"]["atomic_energy"]).abs().max() < 1e-9


def test_cell_change_with_the_same_edge_index_tensor_is_seen():
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    pos, cell, ei, shift, types = _molecule()
    d1 = _data(pos, cell, ei, shift, types)
    o1 = m(d1)
    # isotropic strain: positions and cell scaled, the SAME edge_index / cell-shift tensors re-used
    d2 = dict(d1, pos=d1["pos"] * 1.01, cell=d1["cell"] * 1.01)
    o2 = m(d2)
    fresh = model_from_fixture(fx, torch.float64, emu_lib())
    sv2 = d1["edge_cell_shift"] @ d2["cell"]
    e_ref, f_ref = fresh.energy_forces(d2["pos"], fresh.prepare_graph(d2["edge_index"], d2["atom_types"], 24, sv2))
    assert abs(float(o1["total_energy"]) - float(o2["total_energy"])) > 1e-6
    assert (o2["atomic_energy"].reshape(-1) - e_ref).abs().max() < 1e-12
    assert (o2["forces"] - f_ref).abs().max() < 1e-12
    # changed atom types with the same edge list: new structure key
    t3 = d1["atom_types"].clone()
    t3[0] = (t3[0] + 1) % 3
    o3 = m(dict(d1, atom_types=t3))
    e3, _ = fresh.energy_forces(d1["pos"], fresh.prepare_graph(d1["edge_index"], t3, 24, d1["edge_cell_shift"] @ d1["cell"]))
    assert (o3["atomic_energy"].reshape(-1) - e3).abs().max() < 1e-12
    assert (o3["atomic_energy"] - o1["atomic_energy"]).abs().max() > 1e-8
    # in-place edit of the edge list bumps its version
    ei4 = d1["edge_index"].clone()
    d4 = dict(d1, edge_index=ei4)
    m(d4)
    keep = ei4[0] != 5
    ei4_new = ei4[:, keep]
    d5 = dict(d1, edge_index=ei4_new, edge_cell_shift=d1["edge_cell_shift"][keep])
    o5 = m(d5)
    e5, _ = fresh.energy_forces(d1["pos"], fresh.prepare_graph(ei4_new, d1["atom_types"], 24,
                                                               d5["edge_cell_shift"] @ d1["cell"]))
    assert (o5["atomic_energy"].reshape(-1) - e5).abs().max() < 1e-12


def test_batched_periodic_frames_match_single_frames():
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    pos, cell, ei, shift, types = _molecule()
    f0 = _data(pos, cell, ei, shift, types)
    rng = np.random.default_rng(9)
    pos1 = (pos + rng.normal(0, 0.02, pos.shape)) * 1.015
    cell1 = cell * 1.015 + np.array([[0, 0.05, 0], [0, 0, 0], [0.02, 0, 0]])
    f1 = _data(pos1, cell1, ei, shift, types)  # same topology, different geometry and a triclinic cell
    o0, o1 = m(dict(f0)), m(dict(f1))
    n = pos.shape[0]
    batch = torch.cat([torch.zeros(n, dtype=torch.long), torch.ones(n, dtype=torch.long)])
    both = {"pos": torch.cat([f0["pos"], f1["pos"]]), "edge_index": torch.cat([f0["edge_index"], f1["edge_index"] + n], 1),
            "atom_types": torch.cat([f0["atom_types"], f1["atom_types"]]), "cell": torch.stack([f0["cell"], f1["cell"]]),
            "edge_cell_shift": torch.cat([f0["edge_cell_shift"], f1["edge_cell_shift"]]), "batch": batch}
    ob = m(both)
    assert ob["total_energy"].shape == (2, 1) and ob["stress"].shape == (2, 3, 3)
    for k, (a, b) in {"atomic_energy": (o0, o1), "forces": (o0, o1)}.items():
        want = torch.cat([a[k], b[k]])
        assert (ob[k] - want).abs().max() < 1e-10, k
    for k in ("total_energy", "stress", "virial"):
        want = torch.cat([o0[k], o1[k]])
        assert (ob[k] - want).abs().max() < 1e-9 * max(1.0, float(want.abs().max())), k
