"""ORACLE calibration (test infrastructure; runs ONLY in the build container, where /root/reference is mounted).

`bench.py`'s `cpu_baseline` times oracle/restatement.py (kind "port": the reference cannot travel to the GPU box).  This
script times the REFERENCE ITSELF -- its files imported verbatim behind the leaf shim -- next to that port on the same
host cores and the same edge chunks, so that the port baseline can be translated into "reference CPU path":

    python -m oracle.time_reference  ->  one JSON line (recorded in BASELINE.md section 3)

The reference's eager Contracter materialises [E,u,9,9,9] (186.6 KB/edge at l_max = 2, u = 64: _contract.py:236-241), so
it is evaluated in contiguous center-atom chunks (exact by strict locality, tests/model/test_allegro.py:68-70)."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_loader import import_reference  # noqa: E402
from allegro_amd import graph as G  # noqa: E402


def main(chunk_edges=2800, nchunks=4, reps=3, threads=None):
    import bench
    from oracle import restatement as R

    import_reference()
    from allegro.model import AllegroModel
    from nequip.data import AtomicDataDict as ADD

    threads = threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    g = G.make_si_graph(5)  # 1000 atoms / 28 000 edges of the C3 / C4 lattice
    cfg = bench.si_model_cfg(g.num_edges / g.num_atoms)
    cfg["model_dtype"] = "float32"
    ref_cfg = {k: v for k, v in cfg.items() if k != "model_dtype"}
    model = AllegroModel(model_dtype="float32", **ref_cfg).eval()
    sd = {k[len("func."):]: v.detach() for k, v in model.state_dict().items()}
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], g.num_atoms)
    pos = torch.tensor(g.pos, dtype=torch.float32)
    types = torch.tensor(g.types)
    sv_all = g.shift_vec()
    per_atom = chunk_edges // 28
    chunks = [(i * per_atom, (i + 1) * per_atom) for i in range(nchunks)]

    def run_reference():
        f = torch.zeros_like(pos)
        for a0, a1 in chunks:
            e0, e1 = int(rowptr[a0]), int(rowptr[a1])
            data = {ADD.POSITIONS_KEY: pos, ADD.EDGE_INDEX_KEY: torch.tensor(g.edge_index[:, e0:e1]), ADD.ATOM_TYPE_KEY: types,
                    ADD.CELL_KEY: torch.tensor(g.cell, dtype=torch.float32),
                    ADD.EDGE_CELL_SHIFT_KEY: torch.tensor(g.cell_shift[e0:e1], dtype=torch.float32)}
            f += model(data)["forces"].detach()
        return f

    def run_port():
        f = torch.zeros_like(pos)
        for a0, a1 in chunks:
            e0, e1 = int(rowptr[a0]), int(rowptr[a1])
            out = R.allegro_energy_forces(cfg, sd, pos, torch.tensor(g.edge_index[:, e0:e1]), types,
                                          torch.tensor(sv_all[e0:e1], dtype=torch.float32))
            f += out["forces"]
        return f

    res = {}
    forces = {}
    for name, fn in (("reference_verbatim", run_reference), ("port", run_port)):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            forces[name] = fn()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        E = sum(int(rowptr[a1] - rowptr[a0]) for a0, a1 in chunks)
        res[name] = dict(seconds=t, edges=E, edge_tp_per_s=E * cfg["num_layers"] / t)
    res["max_abs_force_difference"] = float((forces["reference_verbatim"] - forces["port"]).abs().max())
    res["reference_over_port_time"] = res["reference_verbatim"]["seconds"] / res["port"]["seconds"]
    res["threads"] = threads
    res["chunk_edges"] = chunk_edges
    res["note"] = ("C2-C4 model (l_max 2, 2 layers, 64 features) on a 1000-atom Si box, fp32, forward + autograd forces, eager PyTorch CPU; "
                   "the reference's files run verbatim behind oracle/shim")
    return res


if __name__ == "__main__":
    if "--sweep" in sys.argv:
        # the reference / port ratio at every power-of-two thread count of this host (the translation of
        # `cpu_baseline` into "reference CPU path" was only established at 8 threads)
        out = {}
        n = 1
        while n <= (os.cpu_count() or 1):
            r = main(threads=n)
            out[str(n)] = {k: r[k] for k in ("reference_over_port_time", "max_abs_force_difference")}
            out[str(n)].update(reference_s=r["reference_verbatim"]["seconds"], port_s=r["port"]["seconds"])
            n *= 2
        print(json.dumps({"threads": out, "note": "python -m oracle.time_reference --sweep (build container)"}))
    else:
        print(json.dumps(main()))
