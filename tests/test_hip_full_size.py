"""GPU: parity at the sizes the bench reports on (VERDICT r1 "weak" #1).

The full C4 step (97 336 atoms / 2 725 408 edges: the `bench.py` default workload) and the full C5 step (31 944 atoms /
~1.7e6 edges, l_max 3, 3 layers, 128 features, fp64) are evaluated ONCE on the GPU, and the outputs of that run -- not
of a sub-graph run -- are compared with the CPU oracle (oracle/restatement.py) on three contiguous blocks of center
atoms taken at the start, the middle and the end of the CSR, so that large edge offsets, the large-M code paths and the
last partial tiles are all hit.  For a block B the oracle evaluates every edge whose center lies in B+ = B u neighbors(B)
(exact by strict locality, tests/model/test_allegro.py:68-70 of the reference): its E_i are then the model's E_i for
all of B+, and its forces are the model's forces for the atoms of B (every edge touching them is included).
Tolerances: forces 1e-4 eV/A (north star) and energies 5e-5 x scale in fp32; 1e-9 x scale in fp64
(the reference's own model tolerances, tests/model/test_allegro.py:72-74)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _block_check(workload, block_atoms, chunk_edges):
    import bench
    from allegro_amd.nn import HipAllegroModel, PreparedGraph
    from tests.block_utils import oracle_block_check

    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload(workload)
    dtype = {"float32": torch.float32, "float64": torch.float64}[cfg["model_dtype"]]
    model = HipAllegroModel(**cfg).to(dev)
    N = g.num_atoms
    sv_all = g.shift_vec()
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), N,
                          torch.tensor(sv_all, dtype=dtype, device=dev))
    e_full, f_full = model.energy_forces(torch.tensor(g.pos, dtype=dtype, device=dev), graph)
    torch.cuda.synchronize()
    e_full, f_full = e_full.cpu(), f_full.cpu()
    assert torch.isfinite(e_full).all() and torch.isfinite(f_full).all()
    return oracle_block_check(workload, g, cfg, model, e_full, f_full, block_atoms, chunk_edges)


@pytest.mark.parametrize("forward", ["staged", "automatic"])
def test_c4_full_step_vs_oracle_on_three_atom_blocks(forward, monkeypatch):
    # "automatic" is what bench.py and a host without switches run: the fused per-atom-tile forward (28 edges per atom)
    if forward == "automatic":
        monkeypatch.delenv("AA_FUSED", raising=False)
    else:
        monkeypatch.setenv("AA_FUSED", "0")
    _block_check("c4", block_atoms=96, chunk_edges=12000)


def test_c5_full_step_fp64_vs_oracle_on_three_atom_blocks():
    _block_check("c5", block_atoms=48, chunk_edges=2500)


def test_md_loop_conserves_energy_to_second_order_in_dt():
    """NVE loop on the C3 box with the device neighbour list rebuilt EVERY step (`bench.md_loop`): E_pot + E_kin of 10 648 atoms over 60 fs
    moves by a small fraction of the kinetic energy, and by 4x less when the time step is halved -- the signature of a second-order
    integrator driven by forces that ARE the gradient of the energies, across ~70 different neighbour lists, the switch of the
    forward's form when thermal motion pushes a segment past 32 edges, and the workspace / graph-preparation paths of a real host."""
    import bench
    from allegro_amd.nn import HipAllegroModel

    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload("c3")
    model = HipAllegroModel(**cfg).to(dev)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    types = torch.tensor(g.types, device=dev)
    cell = torch.tensor(g.cell, dtype=torch.float64)
    a = bench.md_loop(model, pos, types, cell, float(cfg["r_max"]), steps=60, warmup=0, dt=1.0)
    b = bench.md_loop(model, pos, types, cell, float(cfg["r_max"]), steps=120, warmup=0, dt=0.5)
    print(f"C3 NVE 60 fs: dt 1.0 drift {a['max_abs_drift_eV']:.3f} eV, dt 0.5 drift {b['max_abs_drift_eV']:.3f} eV of "
          f"{a['mean_kinetic_eV']:.0f} eV kinetic; {a['ms_per_md_step_median']:.2f} ms per MD step, max degree {a['max_degree_seen']}")
    assert a["drift_over_mean_kinetic"] < 5e-3 and b["drift_over_mean_kinetic"] < 2e-3
    ratio = a["max_abs_drift_eV"] / max(b["max_abs_drift_eV"], 1e-12)
    assert 2.5 < ratio < 6.0, ratio
    assert a["list_rebuilds"] == 61
