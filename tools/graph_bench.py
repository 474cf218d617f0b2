import sys, time, torch
sys.path.insert(0, '.')
import bench
from allegro_amd.nn import HipAllegroModel, PreparedGraph
dev = torch.device('cuda:0')
for wl in ('c2', 'c3'):
    g, cfg = bench.make_workload(wl)
    m = HipAllegroModel(**cfg).to(dev)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    sv = g.shift_vec()
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), g.num_atoms,
                          torch.tensor(sv, dtype=torch.float32, device=dev) if sv is not None else None)
    for mode in (False, True):
        m.enable_hip_graph(mode)
        for _ in range(20): m.energy_forces(pos, graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): m.energy_forces(pos, graph)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        print(f"{wl} hipGraph={mode}: {dt*1e3:.4f} ms/step")
    m.enable_hip_graph(False)
