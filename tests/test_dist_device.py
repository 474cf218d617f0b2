"""GPU: BASELINE config 3 (atom-block decomposition) with real kernels on `cuda:0` -- results, not timings.

One process, one device: ALL W shards of the C3 box (10 648 atoms, ~2.98e5 edges) are built by the product code
(`HaloShard.from_positions`: slab sort, device cell list of slab + halo, compact local numbering), stepped by the product
kernels, and connected by `allegro_amd.dist.InProcessHaloGroup`, which uses the shards' own pack / accumulate code and plan
tables and replaces only the two `all_to_all_single` by slice copies.  The assembled energies and forces must equal (a) the
one-GPU step on the full graph and (b) the CPU oracle on three atom blocks.  `LocalShard` (replicated positions + one
all-reduce: `north_star`'s wording) is checked the same way with the all-reduce done as an in-process sum.

Two ranks on ONE device (two processes, gloo with host-staged rows -- RCCL refuses two ranks on one GPU): the real
`energy_forces_halo` incl. its stream ordering around both communications, against the reference's golden vectors."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c3(dev):
    import bench
    from allegro_amd.nn import HipAllegroModel, PreparedGraph

    g, cfg = bench.make_workload("c3")
    model = HipAllegroModel(**cfg).to(dev)
    sv = g.shift_vec()
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), g.num_atoms,
                          torch.tensor(sv, dtype=torch.float32, device=dev))
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    e_full, f_full = model.energy_forces(pos, graph)
    torch.cuda.synchronize()
    return g, cfg, model, pos, e_full.clone(), f_full.clone()


@pytest.mark.parametrize("world", [2, 8])
def test_halo_shards_on_device_match_the_one_gpu_step_and_the_oracle(world):
    from allegro_amd.dist import InProcessHaloGroup
    from tests.block_utils import oracle_block_check

    dev = torch.device("cuda:0")
    g, cfg, model, pos, e_full, f_full = _c3(dev)
    N = g.num_atoms
    types = torch.tensor(g.types, device=dev)
    grp = InProcessHaloGroup.from_positions(pos, types, g.cell, cfg["r_max"], world)
    # every rank holds its own slab only; shards partition the atoms; ghost rows have exactly one owner
    assert sum(s.n_own for s in grp.shards) == N
    assert sum(s.n_ghost for s in grp.shards) == sum(sum(s.send_counts) for s in grp.shards) > 0
    assert abs(sum(s.graph.num_edges for s in grp.shards) - g.num_edges) <= 8  # (fp32 vs fp64 classification at r_cut)
    assert max(s.n_own + s.n_ghost for s in grp.shards) < N  # compact: no shard sees the whole frame
    e_all = torch.full((N,), float("nan"), device=dev)
    f_all = torch.full((N, 3), float("nan"), device=dev)
    for rep in range(2):  # twice: persistent buffers are reused, the second step must not see leftovers of the first
        shift = 0.01 if rep == 0 else 0.0
        res = grp.step(model, [pos[s.owned_ids()] + shift for s in grp.shards])
    for s, (e, f) in zip(grp.shards, res):
        e_all[s.owned_ids()] = e
        f_all[s.owned_ids()] = f
    torch.cuda.synchronize()
    assert torch.isfinite(e_all).all() and torch.isfinite(f_all).all()
    dE = float((e_all - e_full).abs().max())
    dF = float((f_all - f_full).abs().max())
    print(f"C3 halo W={world}: vs one-GPU step max|dE_i|={dE:.2e} max|dF|={dF:.2e}; ghosts per shard "
          f"{[s.n_ghost for s in grp.shards]}")
    assert dE <= 5e-5 * max(1.0, float(e_full.abs().max())) and dF <= 2e-5
    oracle_block_check(f"c3 halo W={world}", g, cfg, model, e_all.cpu(), f_all.cpu(), block_atoms=48, chunk_edges=12000)


def test_halo_step_under_hip_graph_replay_matches_eager():
    """The shard step between the two communications replayed from a hipGraph (what bench.py's host-issue analysis times)."""
    from allegro_amd.dist import InProcessHaloGroup

    dev = torch.device("cuda:0")
    g, cfg, model, pos, e_full, f_full = _c3(dev)
    grp = InProcessHaloGroup.from_positions(pos, torch.tensor(g.types, device=dev), g.cell, cfg["r_max"], 8)
    own = [pos[s.owned_ids()] for s in grp.shards]
    eager = [(e.clone(), f.clone()) for e, f in grp.step(model, own)]
    model.enable_hip_graph(True)
    try:
        for _ in range(3):
            replay = grp.step(model, own)
    finally:
        model.enable_hip_graph(False)
    for (e0, f0), (e1, f1) in zip(eager, replay):
        assert torch.equal(e0, e1) and torch.equal(f0, f1)


@pytest.mark.parametrize("world", [2, 8])
def test_local_shards_with_in_process_all_reduce_match_the_one_gpu_step(world):
    from allegro_amd import graph as G
    from allegro_amd.dist import LocalShard

    dev = torch.device("cuda:0")
    g, cfg, model, pos, e_full, f_full = _c3(dev)
    N = g.num_atoms
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], N)
    total = None
    for r in range(world):
        sh = LocalShard(g.edge_index, g.types, N, g.shift_vec(), r, world, dev, torch.float32, rowptr)
        sh.step(model, pos + 0.01)
        sh.step(model, pos)
        total = sh._reduce.clone() if total is None else total + sh._reduce  # the all-reduce of `energy_forces_local`
    e_all, f_all = total.view(-1)[3 * N: 4 * N], total[:N]
    dE = float((e_all - e_full).abs().max())
    dF = float((f_all - f_full).abs().max())
    print(f"C3 LocalShard W={world}: max|dE_i|={dE:.2e} max|dF|={dF:.2e}")
    assert dE <= 5e-5 * max(1.0, float(e_full.abs().max())) and dF <= 2e-5


def _worker_two_ranks_one_device(rank, world, port, q, build="positions"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from allegro_amd.dist import HaloShard, energy_forces_halo
    from tests.golden_utils import load_model_fixture
    from tests.hip_utils import model_from_fixture

    dev = torch.device("cuda:0")
    fx = load_model_fixture("c2", torch.float32)
    m = model_from_fixture(fx, torch.float32, device=dev)
    n = fx["pos"].shape[0]
    cell = np.eye(3) * (2 * 5.431)
    pos = fx["pos"].to(dev)
    if build == "owned":  # a domain-decomposed host: the rank is handed its slab's atoms only (`HaloShard.from_owned`)
        fxx = torch.remainder(pos[:, 0].double() / (2 * 5.431), 1.0)
        mine = torch.nonzero((fxx >= rank / world) & (fxx < (rank + 1) / world)).reshape(-1)
        sh = HaloShard.from_owned(pos[mine].contiguous(), fx["types"].to(dev)[mine], cell, float(fx["cfg"]["r_max"]), rank, world)
        ids = mine
    else:
        sh = HaloShard.from_positions(pos, fx["types"].to(dev), cell, float(fx["cfg"]["r_max"]), rank, world)
        ids = sh.owned_ids()
    assert sh.connected and sh.host_staged
    pos_own = pos[ids].contiguous()
    for k in range(6):  # back-to-back steps with alternating positions: a missing wait around a communication shows up here
        e, f = energy_forces_halo(m, pos_own + (0.01 if k % 2 == 0 else 0.0), sh)
    torch.cuda.synchronize()
    parts = [None] * world
    dist.all_gather_object(parts, (ids.cpu(), e.cpu(), f.cpu(), sh.n_ghost, sum(sh.send_counts)))
    if rank == 0:
        e_all, f_all = torch.zeros(n), torch.zeros(n, 3)
        for ids, ee, ff, _, _ in parts:
            e_all[ids] = ee
            f_all[ids] = ff
        ref = fx["out"]
        q.put(((e_all - ref["atomic_energy"].reshape(-1)).abs().max().item(), (f_all - ref["forces"]).abs().max().item(),
               [(p[3], p[4]) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("build", ["positions", "owned"])
def test_two_ranks_on_one_device_through_energy_forces_halo(build):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() + (0 if build == "positions" else 977)) % 2000
    procs = [ctx.Process(target=_worker_two_ranks_one_device, args=(r, 2, port, q, build)) for r in range(2)]
    for p in procs:
        p.start()
    de, df, stats = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    print(f"2 ranks on cuda:0 (gloo, host-staged rows; shards from {build}): max|dE_i|={de:.2e} max|dF|={df:.2e} vs golden; (ghosts, sent) {stats}")
    assert de < 5e-5 and df < 5e-5
    assert sum(s[0] for s in stats) == sum(s[1] for s in stats) > 0


def test_halo_shards_with_long_segments_take_the_mixed_forward_and_match():
    """A disordered box (jitter 0.4 A: a few atoms with more than 32 neighbours, most with fewer) runs the MIXED form of the fused
    forward (one-tile kernel over all atoms + team kernel over the long ones) -- in the whole box and in every shard, where the owned
    block is a prefix of the local atoms and the per-atom kernels carry the atom-block hint.  Shards = the one-GPU step = the staged
    pipeline."""
    import bench
    from allegro_amd import graph as G
    from allegro_amd.dist import InProcessHaloGroup
    from allegro_amd.nn import HipAllegroModel, PreparedGraph

    dev = torch.device("cuda:0")
    g = G.make_si_graph(8, r_cut=5.0, jitter=0.4, seed=4)  # 3 of 4096 atoms with 33 neighbours, the rest 18..32
    cfg = bench.si_model_cfg(g.num_edges / g.num_atoms)
    cfg["model_dtype"] = "float32"
    model = HipAllegroModel(**cfg).to(dev)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    types = torch.tensor(g.types, device=dev)
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), types, g.num_atoms, torch.tensor(g.shift_vec(), dtype=torch.float32, device=dev))
    assert 32 < graph.max_degree <= 128 and 0.6 <= g.num_edges / (32.0 * g.num_atoms) <= 1.15
    names = [s[0] for s in bench.profile_stages(model, pos, graph, reps=1)]
    assert "fused_fwd" in names, names
    e_full, f_full = (t.clone() for t in model.energy_forces(pos, graph))
    model.check()
    os.environ["AA_FUSED"] = "0"
    try:
        staged = HipAllegroModel(**cfg).to(dev)
        staged.load_state_dict(model.state_dict())
        e_st, f_st = staged.energy_forces(pos, graph)
    finally:
        os.environ.pop("AA_FUSED")
    assert (e_full - e_st).abs().max().item() <= 2e-5 * max(1.0, float(e_st.abs().max())) and (f_full - f_st).abs().max().item() <= 2e-5 * max(1.0, float(f_st.abs().max()))
    grp = InProcessHaloGroup.from_positions(pos, types, g.cell, cfg["r_max"], 2)
    assert any(s.graph.max_degree > 32 for s in grp.shards)
    res = grp.step(model, [pos[s.owned_ids()] for s in grp.shards])
    model.check()
    e_all, f_all = torch.empty_like(e_full), torch.empty_like(f_full)
    for s, (e, f) in zip(grp.shards, res):
        e_all[s.owned_ids()] = e
        f_all[s.owned_ids()] = f
    dE, dF = float((e_all - e_full).abs().max()), float((f_all - f_full).abs().max())
    print(f"disordered Si (max degree {graph.max_degree}): shards vs whole box max|dE_i|={dE:.2e} max|dF|={dF:.2e}")
    assert dE <= 5e-5 * max(1.0, float(e_full.abs().max())) and dF <= 2e-5 * max(1.0, float(f_full.abs().max()))
