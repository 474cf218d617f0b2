"""`pair_allegro` tensor contract on data produced by the REFERENCE's own transform.

`tests/golden/model_c2_ghost.npz` was made by `oracle/make_golden.py: dump_ghost`: the reference's
`allegro_data_settings` (allegro/_compile.py:17-65) applied to the C2 frame -- ghost atoms appended, no cell, edges in
the order it emits them (inside-cell edges first, then the outside-cell ones: NOT sorted by center, :47-58) -- and the
reference model evaluated on the result.  The reference's own test of that transform is
tests/utils/test_compile_utils.py:7-18 (edge lengths are preserved); here its outputs are the known answers for the
`[pos, edge_index, atom_types] -> LMP_OUTPUTS` contract (_compile.py:10-14,68-74): per-atom energies, forces INCLUDING
the ghost rows LAMMPS reverse-communicates, and their sum folded back onto the periodic frame.

CPU: the model through the emulated kernels.  GPU: `ExportableAllegro` (the C++-registered whole-step op) directly,
re-loaded from a saved exported program, and through AOTInductor (`aoti_compile_and_package` -> `aoti_load_package`),
which is the package format `nequip-compile --mode aotinductor` hands to LAMMPS."""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden_utils import GOLDEN_DIR, load_model_fixture
from tests.hip_utils import emu_lib, model_from_fixture

TOL = {torch.float64: 1e-9, torch.float32: 5e-5}  # tests/model/test_allegro.py:72-74


def load_ghost_fixture(dtype):
    z = np.load(os.path.join(GOLDEN_DIR, "model_c2_ghost.npz"))
    base = load_model_fixture(str(z["weights_of"]), dtype)
    assert json.loads(str(z["cfg_json"])) == base["cfg"]
    tag = "out64/" if dtype == torch.float64 else "out32/"
    return dict(base=base, n_local=int(z["n_local"]), pos=torch.tensor(z["pos"]).to(dtype), edge_index=torch.tensor(z["edge_index"]),
                types=torch.tensor(z["types"]), out={k[len(tag):]: torch.tensor(z[k]) for k in z.files if k.startswith(tag)})


def _check(gx, dtype, e_atom, forces, what):
    ref = gx["out"]
    n = gx["n_local"]
    e_atom, forces = e_atom.detach().cpu().reshape(-1), forces.detach().cpu()
    for got, want, name in ((e_atom, ref["atomic_energy"].reshape(-1), "E_i"), (forces, ref["forces"], "F incl. ghost rows")):
        err = (got - want).abs().max().item()
        assert err <= TOL[dtype] * max(1.0, float(want.abs().max())), f"{what} {name}: {err:.3e}"
    # ghosts are never centers: their energies vanish; folded back, the forces are those of the periodic frame
    assert float(e_atom[n:].abs().max()) == 0.0
    ei = gx["edge_index"]
    ghost = ei[1] >= n
    src = gx["base"]["edge_index"]
    # the reference keeps the edge order within the inside / outside groups, so ghost g is the g-th outside-cell edge
    outside = torch.tensor(np.abs(gx["base"]["shift_vec"].numpy()).sum(-1) > 1e-9)
    assert int(outside.sum()) == int(ghost.sum()) == forces.shape[0] - n
    folded = forces[:n].clone().index_add_(0, src[1][outside], forces[n:])
    want = gx["base"]["out"]["forces"]
    assert (folded - want).abs().max().item() <= 4 * TOL[dtype] * max(1.0, float(want.abs().max())), what


def test_fixture_follows_the_reference_transform():
    gx = load_ghost_fixture(torch.float64)
    ei, n = gx["edge_index"], gx["n_local"]
    c = ei[0]
    assert bool((c[1:] < c[:-1]).any()), "inside-cell edges first, then outside-cell: not center-sorted"
    assert int(c.max()) < n  # every center is a local atom
    ghost = ei[1] >= n
    first_ghost = int(torch.nonzero(ghost)[0])
    assert bool(ghost[first_ghost:].all()) and not bool(ghost[:first_ghost].any())
    assert torch.equal(ei[1][ghost], torch.arange(n, gx["pos"].shape[0]))  # one ghost per outside-cell edge, in order
    # tests/utils/test_compile_utils.py:7-18: edge lengths are those of the periodic frame, inside group then outside group
    b = gx["base"]
    r = b["pos"][b["edge_index"][1]] - b["pos"][b["edge_index"][0]] + b["shift_vec"]
    outside = r.new_tensor(np.abs(b["shift_vec"].numpy()).sum(-1) > 1e-9).bool()
    want = torch.cat([r[~outside].norm(dim=1), r[outside].norm(dim=1)])
    got = (gx["pos"][ei[1]] - gx["pos"][ei[0]]).norm(dim=1)
    assert torch.allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_model_on_reference_ghost_data_emulated(dtype):
    gx = load_ghost_fixture(dtype)
    m = model_from_fixture(gx["base"], dtype, emu_lib())
    g = m.prepare_graph(gx["edge_index"], gx["types"], gx["pos"].shape[0], None)
    assert g.perm is not None  # the unsorted list was sorted internally
    e, f = m.energy_forces(gx["pos"], g)
    _check(gx, dtype, e, f, "emulated")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_native_op_on_reference_ghost_data(dtype, forward_mode):
    from allegro_amd.export import ExportableAllegro

    dev = torch.device("cuda:0")
    gx = load_ghost_fixture(dtype)
    m = model_from_fixture(gx["base"], dtype, device=dev)
    ex = ExportableAllegro(m, dev)
    args = (gx["pos"].to(dev), gx["edge_index"].to(dev), gx["types"].to(dev))
    e_atom, e_tot, f, vir = ex(*args)
    _check(gx, dtype, e_atom, f, "native op")
    assert abs(float(e_tot) - float(gx["out"]["atomic_energy"].sum())) <= 10 * TOL[dtype] * abs(float(gx["out"]["atomic_energy"].sum()))
    # the same arguments again (an MD loop holding its list: the op's graph cache) and after an in-place position update
    e2, _, f2, _ = ex(*args)
    assert torch.equal(e2, e_atom) and torch.equal(f2, f)
    # the Python model on the same data
    g = m.prepare_graph(args[1], args[2], args[0].shape[0], None)
    e3, f3 = m.energy_forces(args[0], g)
    _check(gx, dtype, e3, f3, "model")


@pytest.mark.gpu
def test_aotinductor_package_round_trip_on_reference_ghost_data(tmp_path):
    """`nequip-compile --mode aotinductor` -> `pair_allegro`: the exported whole-step program compiled and packaged by
    AOTInductor, loaded back through the AOTI runner (no Python model behind it: config words + weight blob are
    constants of the package, the op comes from liballegro_amd_torch.so)."""
    from allegro_amd.export import ExportableAllegro

    dev = torch.device("cuda:0")
    dtype = torch.float32
    gx = load_ghost_fixture(dtype)
    m = model_from_fixture(gx["base"], dtype, device=dev)
    ex = ExportableAllegro(m, dev)
    args = (gx["pos"].to(dev), gx["edge_index"].to(dev), gx["types"].to(dev))
    ep = torch.export.export(ex, args)
    path = str(tmp_path / "allegro_mi355x.pt2")
    torch._inductor.aoti_compile_and_package(ep, package_path=path)
    assert os.path.getsize(path) > 0
    runner = torch._inductor.aoti_load_package(path)
    got = runner(*args)
    _check(gx, dtype, got[0], got[2], "AOTInductor package")
    want = ex(*args)
    assert torch.equal(got[2], want[2]) and torch.equal(got[3], want[3])


@pytest.mark.gpu
def test_native_op_notices_a_list_rewritten_behind_the_same_tensors():
    """the op caches the CSR of a list by tensor identity + version, which a C++ MD host that refills a
    persistent edge_index through data_ptr() never changes.  Every cache hit is therefore validated by content on the
    device (aa_graph_fingerprint): the call after a raw rewrite returns NaN -- never numbers computed on the stale CSR --
    and the one after that raises."""
    import ctypes

    from allegro_amd.export import ExportableAllegro

    dev = torch.device("cuda:0")
    dtype = torch.float32
    gx = load_ghost_fixture(dtype)
    m = model_from_fixture(gx["base"], dtype, device=dev)
    ex = ExportableAllegro(m, dev)
    pos, ei, types = gx["pos"].to(dev), gx["edge_index"].to(dev).contiguous(), gx["types"].to(dev)
    e0, _, f0, _ = ex(pos, ei, types)
    e1, _, f1, _ = ex(pos, ei, types)  # hit, contents unchanged
    assert torch.equal(e0, e1) and torch.equal(f0, f1)
    # rewrite the contents WITHOUT touching the version counter: reverse the edge order through a raw device copy
    rev = ei.flip(1).contiguous()
    ver = ei._version
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    torch.cuda.synchronize()
    assert hip.hipMemcpy(ei.data_ptr(), rev.data_ptr(), ei.numel() * 8, 3) == 0  # hipMemcpyDeviceToDevice
    assert ei._version == ver
    e2, _, f2, _ = ex(pos, ei, types)
    assert torch.isnan(e2).all() and torch.isnan(f2).all()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="contents of edge_index"):
        ex(pos, ei, types)
    # the entry is gone: the same (rewritten) tensors are now a new list, evaluated correctly (same edges, other order)
    e3, _, f3, _ = ex(pos, ei, types)
    _check(dict(gx, edge_index=ei.cpu()), dtype, e3, f3, "rebuilt after rewrite")


@pytest.mark.gpu
def test_native_op_rejects_packages_of_another_format():
    """a config written by an earlier blob format must fail loudly, not be read with the new offsets."""
    from allegro_amd.export import ExportableAllegro

    dev = torch.device("cuda:0")
    gx = load_ghost_fixture(torch.float32)
    m = model_from_fixture(gx["base"], torch.float32, device=dev)
    ex = ExportableAllegro(m, dev)
    args = (gx["pos"].to(dev), gx["edge_index"].to(dev), gx["types"].to(dev))
    old = list(ex.config)
    old[0] = 0x414C4C4547524F31  # "ALLEGRO1"
    with pytest.raises(RuntimeError, match="another version"):
        torch.ops.allegro_amd_native.energy_forces(*args, None, old, ex.weights)
    nodigest = list(ex.config)
    nodigest[29] = 0
    with pytest.raises(RuntimeError, match="digest"):
        torch.ops.allegro_amd_native.energy_forces(*args, None, nodigest, ex.weights)
