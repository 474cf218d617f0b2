"""Probe (round 5): does running the step of two (or W) atom blocks CONCURRENTLY on separate HIP streams beat one launch sequence over
the whole box?  The step's kernels alternate between matrix-pipe / HBM-bound (chains, fused forward) and vector-issue-bound (moments
reverse) -- co-resident waves of different kernels could fill each other's idle pipes.  One model instance (plan + workspace) per
stream, the W slab shards of the C4 box (HaloShard, no communication), all streams released together.

    python tools/overlap_probe.py [W ...]      -> one line per W: sequential sum, concurrent wall time, whole-box time
"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from allegro_amd import graph as G  # noqa: E402
from allegro_amd.dist import HaloShard  # noqa: E402
from allegro_amd.nn import HipAllegroModel, PreparedGraph  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    pos_np, cell = G.diamond_si(23)
    cfg = bench.si_model_cfg(28.0)
    cfg["model_dtype"] = "float32"
    pos = torch.tensor(pos_np, dtype=torch.float32, device=dev)
    types = torch.zeros(pos.shape[0], dtype=torch.int64, device=dev)
    g, _ = bench.make_workload("c4")
    whole = HipAllegroModel(**cfg).to(dev)
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), types, g.num_atoms, torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev))

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_whole = timeit(lambda: whole.energy_forces(pos, graph))
    print(f"whole box: {t_whole:.3f} ms", flush=True)
    for W in [int(a) for a in sys.argv[1:]] or [2, 4]:
        shards = [HaloShard.from_positions(pos, types, cell, 5.0, r, W, connect=False) for r in range(W)]
        models = [HipAllegroModel(**cfg).to(dev) for _ in range(W)]
        pls = [s.fill_local_positions(pos).clone() for s in shards]
        streams = [torch.cuda.Stream(dev) for _ in range(W)]
        for m, s, p in zip(models, shards, pls):
            m.energy_forces(p, s.graph)
        torch.cuda.synchronize()

        def seq():
            for m, s, p in zip(models, shards, pls):
                m.energy_forces(p, s.graph)

        def conc():
            cur = torch.cuda.current_stream(dev)
            for st, m, s, p in zip(streams, models, shards, pls):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    m.energy_forces(p, s.graph)
            for st in streams:
                cur.wait_stream(st)

        t_seq, t_conc = timeit(seq), timeit(conc)
        print(f"W={W}: shards one after the other {t_seq:.3f} ms, on {W} streams at once {t_conc:.3f} ms, whole box {t_whole:.3f} ms "
              f"(concurrent / whole = {t_conc / t_whole:.3f})", flush=True)
        del shards, models, pls


if __name__ == "__main__":
    main()
