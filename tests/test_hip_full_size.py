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
