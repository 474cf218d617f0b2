"""Probe (round 6): ONE frame cut into W blocks of center atoms (edge slices of the same center-sorted list, the atom-block hints
of aa_graph), stepped on two streams so that block i's REVERSE pass runs beside block i+1's FORWARD (the forward is issue- /
matrix-bound at two waves per SIMD, the reverse pass streams HBM).  aa_model_plan_set_forward_events chains the forwards.

    python tools/pipeline_probe.py [W ...]   -> per W: sequential blocks, pipelined blocks, whole box (ms)
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from allegro_amd import graph as G  # noqa: E402
from allegro_amd.nn import HipAllegroModel, PreparedGraph  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "c4")
    Ws = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2, 4, 8]
    N = g.num_atoms
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    types = torch.tensor(g.types, device=dev)
    ei = torch.tensor(g.edge_index, device=dev)
    sv = torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev)
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], N)
    whole = HipAllegroModel(**cfg).to(dev)
    graph = PreparedGraph(ei, types, N, sv)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    e_ref, f_ref = whole.energy_forces(pos, graph)
    e_ref, f_ref = e_ref.clone(), f_ref.clone()
    t_whole = timeit(lambda: whole.energy_forces(pos, graph))
    print(f"whole box: {t_whole:.3f} ms", flush=True)
    for W in Ws:
        cuts = [0] + [int(np.searchsorted(rowptr, rowptr[-1] * k / W, side="left")) for k in range(1, W)] + [N]
        graphs, models = [], []
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            lo, hi = int(rowptr[a0]), int(rowptr[a1])
            graphs.append(PreparedGraph(ei[:, lo:hi], types, N, sv[lo:hi]))
            models.append(HipAllegroModel(**cfg).to(dev))
            models[-1].load_state_dict(whole.state_dict())
        streams = [torch.cuda.Stream(dev) for _ in range(2)]
        evs = [torch.cuda.Event() for _ in range(W)]
        outs = [m.energy_forces(pos, gr) for m, gr in zip(models, graphs)]
        torch.cuda.synchronize()
        f_sum = sum(o[1] for o in outs)
        err = float((f_sum - f_ref).abs().max())

        def seq():
            for m, gr in zip(models, graphs):
                m._get_lib().check(m._get_lib().lib.aa_model_plan_set_forward_events(m._plan_handle, None, None), "ev")
                m.energy_forces(pos, gr)

        def pipe():
            cur = torch.cuda.current_stream(dev)
            for st in streams:
                st.wait_stream(cur)
            for i, (m, gr) in enumerate(zip(models, graphs)):
                st = streams[i % 2]
                with torch.cuda.stream(st):
                    wait = evs[i - 1].cuda_event if i > 0 else None
                    # (events must exist before their handle is taken: record once up front)
                    m._get_lib().check(m._get_lib().lib.aa_model_plan_set_forward_events(m._plan_handle, wait, evs[i].cuda_event), "ev")
                    m.energy_forces(pos, gr)
            for st in streams:
                cur.wait_stream(st)

        for e in evs:
            e.record()
        torch.cuda.synchronize()
        t_seq = timeit(seq)
        t_pipe = timeit(pipe)
        print(f"W={W}: blocks one after the other {t_seq:.3f} ms, pipelined on 2 streams {t_pipe:.3f} ms, whole box {t_whole:.3f} ms "
              f"(pipelined / whole = {t_pipe / t_whole:.3f}); max |sum of block forces - whole| = {err:.2e}", flush=True)
        seq()
        torch.cuda.synchronize()
        del graphs, models


if __name__ == "__main__":
    main()
