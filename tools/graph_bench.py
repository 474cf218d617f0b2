"""hipGraph replay of a step (aa_model_plan_enable_graph) vs eager launches over box sizes, with the automatic
fused-forward selection on (default) and off (AA_FUSED=0).  Run on the GPU box:  python tools/graph_bench.py"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, ".")


def run(cells):
    import bench
    from allegro_amd.nn import HipAllegroModel, PreparedGraph

    os.environ["AA_BENCH_CELLS"] = str(cells)
    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload("c2")
    m = HipAllegroModel(**cfg).to(dev)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    sv = g.shift_vec()
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), g.num_atoms,
                          torch.tensor(sv, dtype=torch.float32, device=dev) if sv is not None else None)
    out = []
    for mode in (False, True):
        m.enable_hip_graph(mode)
        for _ in range(30):
            m.energy_forces(pos, graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            m.energy_forces(pos, graph)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 300 * 1e3)
    m.enable_hip_graph(False)
    print(f"cells={cells} atoms={g.num_atoms} AA_FUSED={os.environ.get('AA_FUSED', 'auto')}: eager {out[0]:.4f}  hipGraph {out[1]:.4f} ms/step", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:
        for fused in (None, "0"):
            env = dict(os.environ)
            env.pop("AA_FUSED", None)
            if fused is not None:
                env["AA_FUSED"] = fused
            for cells in (2, 4, 5, 8):
                subprocess.run([sys.executable, __file__, str(cells)], env=env, check=False)
