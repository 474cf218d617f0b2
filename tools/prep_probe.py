import sys, time
sys.path.insert(0, ".")
import torch, bench
from allegro_amd.nn import neighbor_list
dev = torch.device("cuda:0")
for wl in ("c3", "c4"):
    g, cfg = bench.make_workload(wl)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    types = torch.tensor(g.types, device=dev)
    cell = torch.tensor(g.cell, dtype=torch.float64)
    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    nl = neighbor_list(pos, cell, True, 5.0)
    print(wl, "neighbor_list", round(t(lambda: neighbor_list(pos, cell, True, 5.0)), 3), "prepare", round(t(lambda: nl.prepare(types)), 3),
          "prepare(no transpose)", round(t(lambda: nl.prepare(types, transposed=False)), 3),
          "argsort", round(t(lambda: torch.argsort(nl.edge_index[1], stable=True)), 3),
          "argsort+to32", round(t(lambda: torch.argsort(nl.edge_index[1], stable=True).to(torch.int32)), 3),
          "bincount+cumsum", round(t(lambda: torch.cumsum(torch.bincount(nl.edge_index[1], minlength=g.num_atoms), 0)), 3))
