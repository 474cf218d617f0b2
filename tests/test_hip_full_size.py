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
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _block_check(workload, block_atoms, chunk_edges):
    import bench
    from allegro_amd import graph as G
    from allegro_amd.nn import HipAllegroModel, PreparedGraph
    from oracle import restatement as R

    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload(workload)
    dtype = {"float32": torch.float32, "float64": torch.float64}[cfg["model_dtype"]]
    model = HipAllegroModel(**cfg).to(dev)
    N, E = g.num_atoms, g.num_edges
    sv_all = g.shift_vec()
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), N,
                          torch.tensor(sv_all, dtype=dtype, device=dev))
    e_full, f_full = model.energy_forces(torch.tensor(g.pos, dtype=dtype, device=dev), graph)
    torch.cuda.synchronize()
    e_full, f_full = e_full.cpu(), f_full.cpu()
    assert torch.isfinite(e_full).all() and torch.isfinite(f_full).all()
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], N)
    sd = {k[len("func."):]: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    center, nbr = g.edge_index[0], g.edge_index[1]
    worst = dict(dE=0.0, dF=0.0)
    for b0 in (0, N // 2 - block_atoms // 2, N - block_atoms):
        B = np.arange(b0, b0 + block_atoms)
        Bp = np.unique(np.concatenate([B, nbr[rowptr[b0]:rowptr[b0 + block_atoms]]]))  # B+ (symmetric neighbor list)
        eids = np.concatenate([np.arange(rowptr[a], rowptr[a + 1]) for a in Bp])       # center-sorted (Bp is sorted)
        atoms = np.unique(np.concatenate([Bp, nbr[eids]]))
        loc = -np.ones(N, dtype=np.int64)
        loc[atoms] = np.arange(atoms.size)
        ei_loc = torch.tensor(np.stack([loc[center[eids]], loc[nbr[eids]]]))
        out = R.allegro_energy_forces_chunked(dict(cfg), sd, torch.tensor(g.pos[atoms], dtype=dtype), ei_loc,
                                              torch.tensor(g.types[atoms]), torch.tensor(sv_all[eids], dtype=dtype),
                                              chunk_edges)
        e_o = out["atomic_energy"].reshape(-1)[loc[Bp]]
        f_o = out["forces"][loc[B]]
        dE = float((e_full[Bp] - e_o).abs().max())
        dF = float((f_full[B] - f_o).abs().max())
        tol_e = (5e-5 if dtype == torch.float32 else 1e-9) * max(1.0, float(e_o.abs().max()))
        tol_f = 1e-4 if dtype == torch.float32 else 1e-9 * max(1.0, float(f_o.abs().max()))
        print(f"{workload} block@{b0}: |B+|={Bp.size} edges={eids.size} max|dE|={dE:.3e} (tol {tol_e:.1e}) "
              f"max|dF|={dF:.3e} (tol {tol_f:.1e})")
        assert dE <= tol_e and dF <= tol_f, (workload, b0, dE, tol_e, dF, tol_f)
        worst = dict(dE=max(worst["dE"], dE), dF=max(worst["dF"], dF))
    return worst


@pytest.mark.parametrize("forward", ["staged", "automatic"])
def test_c4_full_step_vs_oracle_on_three_atom_blocks(forward, monkeypatch):
    # "automatic" is what bench.py and a host without switches run: the fused per-atom-tile forward (28 edges per atom)
    if forward == "automatic":
        monkeypatch.delenv("AA_FUSED", raising=False)
    else:
        monkeypatch.setenv("AA_FUSED", "0")
    _block_check("c4", block_atoms=96, chunk_edges=12000)


def test_c5_full_step_fp64_vs_oracle_on_three_atom_blocks():
    _block_check("c5", block_atoms=6, chunk_edges=2500)
