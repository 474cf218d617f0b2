"""On-device neighbor list (aa_nl_count / aa_nl_fill) against a brute-force image enumeration in numpy float64.
CPU: the kernels run in the test-only emulation build; `-m gpu`: the same cases on the gfx950 library plus a
C3-sized box against the host cell list."""
import itertools

import numpy as np
import pytest
import torch

from tests.hip_utils import emu_lib
from allegro_amd.nn import neighbor_list


def brute_force(pos, cell, pbc, r_cut):
    """All (i, j, S) with |pos[j] - pos[i] + S @ cell| < r_cut, (i == j only for S != 0)."""
    pos, cell = np.asarray(pos, np.float64), np.asarray(cell, np.float64)
    inv = np.linalg.inv(cell)
    h = 1.0 / np.linalg.norm(inv, axis=0)
    frac = pos @ inv
    span = np.ceil(frac.max(0) - frac.min(0)).astype(int) if len(pos) else np.zeros(3, int)
    reps = [range(-(int(np.ceil(r_cut / h[a])) + span[a]), int(np.ceil(r_cut / h[a])) + span[a] + 1) if pbc[a] else [0]
            for a in range(3)]
    out = set()
    for S in itertools.product(*reps):
        d = pos[None, :, :] + (np.array(S) @ cell)[None, None, :] - pos[:, None, :]
        ii, jj = np.nonzero((d ** 2).sum(-1) < r_cut * r_cut)
        for i, j in zip(ii, jj):
            if i != j or any(S):
                out.add((int(i), int(j)) + tuple(int(s) for s in S))
    return out


def as_set(nl):
    ei, cs = nl.edge_index.cpu().numpy(), nl.cell_shift.cpu().numpy()
    return [(int(ei[0, e]), int(ei[1, e])) + tuple(int(s) for s in cs[e]) for e in range(ei.shape[1])]


def check(pos, cell, pbc, r_cut, dtype, lib, device="cpu"):
    p = torch.tensor(pos, dtype=dtype, device=device)
    nl = neighbor_list(p, cell, pbc, r_cut, lib=lib)
    got = as_set(nl)
    assert len(got) == len(set(got)), "duplicate edges"
    want = brute_force(p.double().cpu().numpy(), cell, pbc, r_cut)
    assert set(got) == want, (len(got), len(want))
    ei = nl.edge_index.cpu().long()
    rp = nl.rowptr.cpu().long()
    assert bool((ei[0, 1:] >= ei[0, :-1]).all()) if ei.shape[1] > 1 else True
    assert torch.equal(rp[1:] - rp[:-1], torch.bincount(ei[0], minlength=p.shape[0]))
    sv = nl.cell_shift.cpu().double() @ torch.tensor(cell, dtype=torch.float64)
    assert (nl.shift_vec.cpu().double() - sv).abs().max().item() <= (1e-6 if dtype == torch.float32 else 1e-12) * \
        max(1.0, float(sv.abs().max())) if ei.shape[1] else True
    return nl


CASES = {
    "orthorhombic": (lambda r: r.uniform(0, 1, (60, 3)) * [13.0, 11.0, 12.0], np.diag([13.0, 11.0, 12.0]), (1, 1, 1), 3.1),
    "small_box_many_images": (lambda r: r.uniform(0, 1, (5, 3)) * [2.2, 3.0, 7.5], np.diag([2.2, 3.0, 7.5]), (1, 1, 1), 3.4),
    "triclinic": (lambda r: r.uniform(-0.3, 1.4, (40, 3)) @ np.array([[9.0, 0, 0], [2.5, 8.0, 0], [-1.5, 2.0, 7.0]]),
                  np.array([[9.0, 0, 0], [2.5, 8.0, 0], [-1.5, 2.0, 7.0]]), (1, 1, 1), 3.0),
    "slab_with_atoms_outside": (lambda r: r.uniform(-0.4, 1.5, (50, 3)) * [8.0, 9.0, 14.0], np.diag([8.0, 9.0, 14.0]),
                                (1, 1, 0), 3.2),
    "molecule_no_pbc": (lambda r: r.uniform(0, 6.0, (20, 3)), np.diag([6.0, 6.0, 6.0]), (0, 0, 0), 2.5),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_device_neighbor_list_matches_brute_force(name, dtype):
    gen, cell, pbc, rc = CASES[name]
    pos = gen(np.random.default_rng(7))
    check(pos, cell, pbc, rc, dtype, emu_lib())


def test_no_atoms_and_isolated_atoms():
    lib = emu_lib()
    nl = neighbor_list(torch.zeros((0, 3), dtype=torch.float64), np.eye(3) * 5, True, 2.0, lib=lib)
    assert nl.num_edges == 0 and nl.rowptr.tolist() == [0]
    nl = neighbor_list(torch.tensor([[0.0, 0, 0], [10.0, 10, 10]], dtype=torch.float64), np.eye(3) * 40, True, 2.0, lib=lib)
    assert nl.num_edges == 0 and nl.rowptr.tolist() == [0, 0, 0]


def test_model_on_device_list_equals_model_on_host_list():
    """Same energies/forces from the device-built graph and from the host list the fixtures were made with."""
    from tests.golden_utils import load_model_fixture
    from tests.hip_utils import fixture_data, model_from_fixture
    from oracle import make_golden as MG

    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    data, sv = fixture_data(fx, torch.float64)
    g = MG.molecule_graph()  # the geometry of the fixture (box 9 A, r_cut 4 A)
    assert np.allclose(g.pos, data["pos"].numpy())
    nl = neighbor_list(data["pos"], g.cell, True, 4.0, lib=emu_lib())
    assert nl.num_edges == data["edge_index"].shape[1]
    e, f = m.energy_forces(data["pos"], nl.prepare(data["atom_types"]))
    ref = fx["out"]
    assert (e - ref["atomic_energy"].reshape(-1)).abs().max().item() < 1e-9
    assert (f - ref["forces"]).abs().max().item() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_neighbor_list_on_gpu(name):
    gen, cell, pbc, rc = CASES[name]
    pos = gen(np.random.default_rng(11))
    for dtype in (torch.float64, torch.float32):
        check(pos, cell, pbc, rc, dtype, None, device="cuda:0")


@pytest.mark.gpu
def test_c3_box_on_gpu_matches_host_cell_list_and_model():
    """10 648-atom Si box (BASELINE config 3): identical edge set to the host list; same model output through it."""
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel
    from oracle import make_golden as MG

    dev = torch.device("cuda:0")
    g = G.make_si_graph(11)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    nl = neighbor_list(pos, g.cell, True, 5.0)
    assert nl.num_edges == g.num_edges == 28 * g.num_atoms
    got = torch.cat((nl.edge_index.long(), nl.cell_shift.long().T), 0).cpu().numpy()
    want = np.concatenate((g.edge_index, g.cell_shift.T.astype(np.int64)), 0)
    assert set(map(tuple, got.T)) == set(map(tuple, want.T))
    nl2 = neighbor_list(pos, g.cell, True, 5.0)
    assert torch.equal(nl.edge_index, nl2.edge_index) and torch.equal(nl.cell_shift, nl2.cell_shift)  # reproducible
    cfg = MG.si_cfg(2, 2, 64)
    cfg["model_dtype"] = "float32"
    m = HipAllegroModel(**cfg).to(dev)
    types = torch.zeros(g.num_atoms, dtype=torch.long, device=dev)
    e1, f1 = m.energy_forces(pos, nl.prepare(types))
    sv = torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev)
    e2, f2 = m.energy_forces(pos, m.prepare_graph(torch.tensor(g.edge_index, device=dev), types, g.num_atoms, sv))
    assert (e1 - e2).abs().max().item() <= 2e-5 * max(1.0, float(e2.abs().max()))
    assert (f1 - f2).abs().max().item() <= 2e-5 * max(1.0, float(f2.abs().max()))


def test_multi_block_scans_against_host_cell_list():
    """> 4096 atoms and > 4096 cells (both prefix sums span several workgroups) vs the host KD-tree list."""
    from allegro_amd import graph as G

    rng = np.random.default_rng(3)
    box, rc = 60.0, 3.0
    pos = rng.uniform(0, box, size=(6000, 3))
    cell = np.eye(3) * box
    ei, shift = G.neighbor_list_pbc(pos, cell, rc)
    nl = neighbor_list(torch.tensor(pos), cell, True, rc, lib=emu_lib())
    assert nl.num_edges == ei.shape[1]
    got = set(as_set(nl))
    want = set((int(ei[0, e]), int(ei[1, e])) + tuple(int(s) for s in shift[e]) for e in range(ei.shape[1]))
    assert got == want
    assert int(nl.rowptr[-1]) == nl.num_edges and bool((nl.rowptr[1:] >= nl.rowptr[:-1]).all())


def test_invalid_inputs_are_reported_not_crashed():
    from allegro_amd._lib import AllegroError

    lib = emu_lib()
    pos = torch.zeros((3, 3), dtype=torch.float64)
    with pytest.raises(AllegroError, match="r_cut"):
        neighbor_list(pos, np.eye(3) * 5, True, -1.0, lib=lib)
    with pytest.raises(AllegroError, match="singular"):
        neighbor_list(pos, np.zeros((3, 3)), True, 2.0, lib=lib)
    with pytest.raises(AllegroError, match="GPU"):
        neighbor_list(pos, np.eye(3) * 5, True, 2.0)  # default (gfx950) library: CPU tensors are refused


def _ghost_layout_case(lib, dev):
    """`DeviceNeighborList.ghost_layout` (device-side ghost construction of the `pair_allegro` contract,
    allegro/_compile.py:28-63) against the host construction of allegro_amd/graph.py, and the model on it: local energies
    equal the periodic evaluation, ghost-row forces folded back onto their source atoms equal the periodic forces."""
    from allegro_amd import graph as G
    from allegro_amd.nn import neighbor_list
    from tests.golden_utils import load_model_fixture
    from tests.hip_utils import model_from_fixture

    fx = load_model_fixture("c2", torch.float64)
    m = model_from_fixture(fx, torch.float64, lib, dev)
    g = G.make_si_graph(2)
    pos = torch.tensor(g.pos, device=dev)
    types = torch.tensor(g.types, device=dev)
    nl = neighbor_list(pos, torch.tensor(g.cell), True, 5.0, lib=lib)
    pos_x, types_x, ei_x, src = nl.ghost_layout(pos, types)
    gg = G.to_ghost_layout(g)
    n = g.num_atoms
    assert pos_x.shape[0] == gg.num_atoms and ei_x.shape[1] == g.num_edges
    assert bool((ei_x[1] >= n).sum() == gg.num_atoms - n)
    # same multiset of (center, neighbor position) pairs as the host construction
    def key(p, ei):
        v = p[ei[1]] - p[ei[0]]
        return torch.sort((ei[0].double() * 1e3 + (v * torch.tensor([1.0, 7.0, 49.0], dtype=torch.float64, device=v.device)).sum(-1)))[0]
    kd = key(pos_x, ei_x).cpu()
    kh = key(torch.tensor(gg.pos), torch.tensor(gg.edge_index))
    assert (kd - kh).abs().max().item() < 1e-9
    e, f = m.energy_forces(pos_x, m.prepare_graph(ei_x, types_x, pos_x.shape[0]))
    folded = f[:n].clone().index_add_(0, src, f[n:]).cpu()
    assert (folded - fx["out"]["forces"]).abs().max().item() < 1e-9
    assert (e[:n].cpu() - fx["out"]["atomic_energy"].reshape(-1)).abs().max().item() < 1e-9


def test_device_ghost_layout_emulated():
    from tests.hip_utils import emu_lib

    _ghost_layout_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_device_ghost_layout_on_gpu():
    _ghost_layout_case(None, torch.device("cuda:0"))


def test_graph_fingerprint_is_position_dependent_and_reproducible_emulated():
    """`aa_graph_fingerprint` (what the exported op validates its graph cache with): same contents -> same 128 bits, any change
    of an entry OR of the order -> different; int32 and int64 type arrays."""
    import ctypes as C

    from tests.hip_utils import emu_lib

    lib = emu_lib().lib
    lib.aa_graph_fingerprint.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    g = torch.Generator().manual_seed(3)
    ei = torch.randint(0, 50, (2, 777), generator=g)
    types = torch.randint(0, 3, (50,), generator=g)

    def fp(e, t):
        out = torch.zeros(2, dtype=torch.int64)
        e = e.contiguous()
        assert lib.aa_graph_fingerprint(e.data_ptr(), e.stride(0), e.shape[1], t.data_ptr(), int(t.dtype == torch.int64), t.numel(),
                                        out.data_ptr(), None) == 0
        return tuple(out.tolist())

    base = fp(ei, types)
    assert fp(ei.clone(), types.clone()) == base
    e2 = ei.clone()
    e2[1, 500] += 1
    assert fp(e2, types)[0] != base[0] and fp(e2, types)[1] == base[1]
    assert fp(ei.flip(1), types)[0] != base[0]  # a permutation of the same edges is another list (other CSR permutation)
    t2 = types.clone()
    t2[7] = (t2[7] + 1) % 3
    assert fp(ei, t2)[1] != base[1] and fp(ei, t2)[0] == base[0]
    assert fp(ei, types.to(torch.int32)) == base  # same values, other integer width
    assert fp(ei[:, :0], types[:0]) == (0, 0)


def _transpose_case(lib, dev):
    """`aa_graph_transpose` (counting sort by atomics + per-atom group sort + the three graph hints on the device) equals the stable
    argsort / bincount / cumsum construction entry by entry, on ragged lists with empty atoms at both ends, repeated pairs and
    no edges at all; `PreparedGraph` takes it whenever a library is at hand."""
    import numpy as np

    from allegro_amd.nn import PreparedGraph

    rng = np.random.default_rng(12)
    for n, e, lo, hi in ((50, 700, 5, 45), (7, 0, 0, 7), (300, 9000, 0, 300), (3, 40, 1, 2), (5, 1500, 0, 5), (40, 6000, 0, 40)):
        center = np.sort(rng.integers(lo, max(hi, lo + 1), size=e)) if e else np.zeros(0, dtype=np.int64)
        nbr = rng.integers(0, n, size=e)
        ei = torch.tensor(np.stack([center, nbr]), dtype=torch.int64, device=dev)
        types = torch.zeros(n, dtype=torch.int64, device=dev)
        a = PreparedGraph(ei, types, n, None, lib=lib)
        assert a.t_perm.dtype == torch.int32 and a.t_rowptr.dtype == torch.int32
        want_perm = torch.argsort(ei[1], stable=True).to(torch.int32)
        want_row = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        want_row[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n), 0).to(torch.int32)
        assert torch.equal(a.t_perm, want_perm) and torch.equal(a.t_rowptr, want_row), (n, e)
        deg = np.bincount(center, minlength=n) if e else np.zeros(n, dtype=np.int64)
        if e:
            assert (a.atom_begin, a.atom_end, a.max_degree) == (int(center[0]), int(center[-1]) + 1, int(deg.max()))
        else:
            assert (a.atom_begin, a.atom_end, a.max_degree) == (0, 0, 0)


def test_graph_transpose_matches_stable_argsort_emulated():
    from tests.hip_utils import emu_lib

    _transpose_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_graph_transpose_matches_stable_argsort_on_gpu():
    _transpose_case(None, torch.device("cuda:0"))
    # C4-sized list: 97 336 atoms, 2.7e6 edges, against the tensor-operation construction
    import bench
    from allegro_amd.nn import neighbor_list

    g, cfg = bench.make_workload("c3")
    dev = torch.device("cuda:0")
    nl = neighbor_list(torch.tensor(g.pos, dtype=torch.float32, device=dev), torch.tensor(g.cell), True, 5.0)
    pg = nl.prepare(torch.zeros(g.num_atoms, dtype=torch.int64, device=dev))
    assert torch.equal(pg.t_perm, torch.argsort(nl.edge_index[1].long(), stable=True).to(torch.int32))
    assert pg.max_degree == int((nl.rowptr[1:] - nl.rowptr[:-1]).max())
