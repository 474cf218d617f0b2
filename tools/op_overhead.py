"""Per-step cost of the AOTInductor / LAMMPS route (the C++-registered op `allegro_amd_native::energy_forces`,
allegro_amd/csrc/torch_ops.cpp) against `HipAllegroModel.energy_forces` on a prepared graph: what the op's own host work
(sortedness check, CSR + transposed CSR, degree reduction, workspace) costs, and what its graph cache recovers when the
caller hands the same neighbour-list tensors again (an MD loop that rebuilds its list every few steps).

    python tools/op_overhead.py  ->  one JSON line (rows: 64 / 1000 / 10 648 atoms)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from allegro_amd import graph as G  # noqa: E402
from allegro_amd.export import ExportableAllegro  # noqa: E402
from allegro_amd.nn import HipAllegroModel  # noqa: E402


def timed(fn, steps, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda:0")
    rows = []
    for cells in (2, 5, 11):
        g = G.make_si_graph(cells)
        cfg = bench.si_model_cfg(g.num_edges / g.num_atoms)
        cfg["model_dtype"] = "float32"
        m = HipAllegroModel(**cfg).to(dev)
        pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
        ei = torch.tensor(g.edge_index, device=dev)
        types = torch.tensor(g.types, device=dev)
        sv = torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev)
        pg = m.prepare_graph(ei, types, g.num_atoms, sv)
        ex = ExportableAllegro(m, dev)
        steps = 300 if cells < 11 else 100
        t_model = timed(lambda: m.energy_forces(pos, pg), steps)
        t_hit = timed(lambda: ex(pos, ei, types, sv), steps)
        t_miss = timed(lambda: ex(pos, ei.clone(), types, sv), steps)  # a new list tensor every step: full host work
        rows.append(dict(atoms=g.num_atoms, edges=g.num_edges, model_prepared_graph_ms=t_model, native_op_same_list_ms=t_hit,
                         native_op_new_list_every_step_ms=t_miss, op_overhead_cached_ms=t_hit - t_model,
                         op_overhead_uncached_ms=t_miss - t_model))
    print(json.dumps({"op_overhead": rows, "note": "fp32 C2-C4 model on Si boxes; the native op also returns the virial (two more small launches)"}))


if __name__ == "__main__":
    main()
