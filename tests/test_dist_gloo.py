"""CPU, world_size 2, gloo: the N>1 path (atom-block sharding + force all-reduce) reproduces the
single-process result.  Kernels run through the test-only emulation build (no GPU here)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from allegro_amd.dist import energy_forces_sharded, local_graph
    from tests.golden_utils import load_model_fixture
    from tests.hip_utils import emu_lib, model_from_fixture

    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, emu_lib())
    sv = fx["shift_vec"].numpy()
    g, owned = local_graph(fx["edge_index"].numpy(), fx["types"].numpy(), fx["pos"].shape[0], sv, rank, world, "cpu",
                           torch.float64)
    e, f = energy_forces_sharded(m, fx["pos"], g, owned)
    if rank == 0:
        q.put(((e - fx["out"]["atomic_energy"].reshape(-1)).abs().max().item(),
               (f - fx["out"]["forces"]).abs().max().item(), g.num_edges))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_reference():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    de, df, e_local = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 0 < e_local < 192  # rank 0 really held only a share of the edges
    assert de < 1e-8 and df < 1e-8
