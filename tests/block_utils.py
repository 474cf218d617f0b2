"""Test infrastructure: outputs of ONE full-size GPU step checked against the CPU oracle on contiguous blocks of center atoms.

For a block B the oracle evaluates every edge whose center lies in B+ = B u neighbors(B) (exact by strict locality,
tests/model/test_allegro.py:68-70 of the reference): its E_i are then the model's E_i for all of B+, and its forces are the
model's forces for the atoms of B (every edge touching them is included).  Tolerances: forces 1e-4 eV/A (north star) and
energies 5e-5 x scale in fp32; 1e-9 x scale in fp64 (the reference's own model tolerances, tests/model/test_allegro.py:72-74)."""
import numpy as np
import torch


def oracle_block_check(label, g, cfg, model, e_full, f_full, block_atoms, chunk_edges, starts=None):
    """`e_full` [N], `f_full` [N,3] (CPU): what the path under test produced for the whole frame `g` (caller's numbering)."""
    from allegro_amd import graph as G
    from oracle import restatement as R

    dtype = {"float32": torch.float32, "float64": torch.float64}[cfg["model_dtype"]]
    N = g.num_atoms
    sv_all = g.shift_vec()
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], N)
    sd = {k[len("func."):]: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    center, nbr = g.edge_index[0], g.edge_index[1]
    worst = dict(dE=0.0, dF=0.0)
    for b0 in (starts if starts is not None else (0, N // 2 - block_atoms // 2, N - block_atoms)):
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
        print(f"{label} block@{b0}: |B+|={Bp.size} edges={eids.size} max|dE|={dE:.3e} (tol {tol_e:.1e}) "
              f"max|dF|={dF:.3e} (tol {tol_f:.1e})")
        assert dE <= tol_e and dF <= tol_f, (label, b0, dE, tol_e, dF, tol_f)
        worst = dict(dE=max(worst["dE"], dE), dF=max(worst["dF"], dF))
    return worst
