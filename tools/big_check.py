"""64-bit indexing check at a size where E * row_width exceeds 2^32 elements: one step on a large Si box vs. the same
step evaluated as 8 independent atom blocks (exact by strict locality; each block's offsets stay below 2^31).
Usage (GPU box): python tools/big_check.py [cells=36]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allegro_amd import graph as G  # noqa: E402
from allegro_amd.dist import partition_atoms  # noqa: E402
from allegro_amd.nn import HipAllegroModel, PreparedGraph  # noqa: E402
import bench  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 36
dev = torch.device("cuda:0")
t0 = time.time()
g = G.make_si_graph(cells)
N, E = g.num_atoms, g.num_edges
print(f"Si {cells}^3: N={N} E={E}  E*768={E * 768:.3e} (2^32={2 ** 32:.3e})  host graph {time.time() - t0:.1f}s", flush=True)
cfg = bench.si_model_cfg()
model = HipAllegroModel(**cfg).to(dev)
model._ensure_plan()
lib = model._get_lib()
need = lib.lib.aa_model_workspace_bytes(model._plan_handle, N, E, 1)
free, total = torch.cuda.mem_get_info(dev)
print(f"workspace {need / 2 ** 30:.1f} GiB, free {free / 2 ** 30:.1f} of {total / 2 ** 30:.1f} GiB", flush=True)
if need > 0.8 * free:
    raise SystemExit("not enough memory for this size; choose fewer cells")
pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
types = torch.tensor(g.types, device=dev)
ei = torch.tensor(g.edge_index, device=dev)
sv = torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev)
full = PreparedGraph(ei, types, N, sv)
e_full, f_full = model.energy_forces(pos, full)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3):
    model.energy_forces(pos, full)
torch.cuda.synchronize()
ms = (time.perf_counter() - t1) / 3 * 1e3
print(f"full step {ms:.1f} ms = {E * 2 / ms * 1e3:.3e} edge-TP/s", flush=True)
e_full, f_full = e_full.clone(), f_full.clone()
rowptr = G.csr_from_sorted_centers(g.edge_index[0], N)
cuts = partition_atoms(rowptr, 8)
e_sum = torch.zeros_like(e_full)
f_sum = torch.zeros_like(f_full)
for r in range(8):
    a0, a1 = cuts[r], cuts[r + 1]
    e0, e1 = int(rowptr[a0]), int(rowptr[a1])
    blk = PreparedGraph(ei[:, e0:e1], types, N, sv[e0:e1])
    e_b, f_b = model.energy_forces(pos, blk)
    e_sum[a0:a1] = e_b[a0:a1]
    f_sum += f_b
torch.cuda.synchronize()
de = float((e_sum - e_full).abs().max())
df = float((f_sum - f_full).abs().max())
print(f"max|dE_i| = {de:.3e} (|E_i|max {float(e_full.abs().max()):.3f})   max|dF| = {df:.3e} (|F|max {float(f_full.abs().max()):.3f})")
assert torch.isfinite(f_full).all() and torch.isfinite(e_full).all()
assert de <= 2e-5 * max(1.0, float(e_full.abs().max())) and df <= 2e-5 * max(1.0, float(f_full.abs().max()))
print("OK: full-size step equals the sum of its 8 atom blocks")
