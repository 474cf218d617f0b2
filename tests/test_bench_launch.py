"""`python bench.py --gpus N` launches itself (one process per rank under torch.distributed.run) and prints ONE JSON line.
Here without a GPU: AA_BENCH_EMULATED=1 = CPU tensors through the emulation build of the kernel sources, gloo.  Also: a rank
handed atoms outside its slab makes EVERY rank of `HaloShard.from_owned` raise (nobody is left blocked in a collective)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_launches_itself_and_prints_one_line():
    from tests.hip_utils import emu_lib

    emu_lib()  # (build the emulation library once, before two ranks race for it)
    env = dict(os.environ, AA_BENCH_EMULATED="1", AA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "c2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"].startswith("edge tensor-products/sec") and d["unit"] == "edge-TP/s"
    assert "x2 over gloo" in d["config"]["parallelism"] and "from_owned" in d["config"]["parallelism"]
    assert d["config"]["atoms"] == 64 and d["config"]["edges"] > 1000
    rk = d["config"]["rank_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and sum(rk["edges_per_rank"]) == d["config"]["edges"]
    assert rk["max"] <= d["ms_per_step"] * 1.05 and "EMULATED" in d["data"]
    assert abs(d["value"] - d["config"]["edges"] * 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def _bad_slab_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from allegro_amd.dist import HaloShard

    cell = torch.eye(3, dtype=torch.float64) * 10.0
    x = torch.rand(6, 3, dtype=torch.float64) * 10.0
    x[:, 0] = (rank + torch.rand(6, dtype=torch.float64) * 0.98 + 0.01) * 10.0 / world
    if rank == 1:
        x[0, 0] = 0.5  # an atom of rank 0's slab
    try:
        HaloShard.from_owned(x, torch.zeros(6, dtype=torch.int64), cell, 3.0, rank, world)
        q.put((rank, "no error"))
    except ValueError as ex:
        q.put((rank, str(ex)))
    dist.barrier()
    dist.destroy_process_group()


def test_from_owned_bad_slab_raises_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35100 + os.getpid() % 1500
    procs = [ctx.Process(target=_bad_slab_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(3))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(3):
        assert "rank(s) [1]" in got[r], got
