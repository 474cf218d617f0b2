"""NVE molecular dynamics on one MI355X with everything on the device: the metric's "ns/day for a 10^5-atom box" measured as a LOOP --
neighbour list (`aa_nl_*`), graph preparation, the hot path (`aa_model_energy_forces`), velocity-Verlet update -- instead of
0.0864 / t_step, and the energy conservation of that trajectory: an end-to-end check at full size that the forces the hot path
returns are the gradient of the energies it returns (a wrong sign, a missing image, a dropped neighbour contribution or a stale
list shows up as a drift of E_pot + E_kin, not as a number in a table).

    python tools/md_loop.py [--workload c4] [--steps 200] [--dt 1.0] [--temperature 300] [--skin 0.0 | 0.4] > profiles/rNN_md_loop_c4.json

`--skin 0` (default): the list is rebuilt EVERY step at r_cut (the device list + graph preparation cost ~1 ms at C4, less than
the ~33 % more edges a 0.5 A skin would make every step compute).  `--skin s`: list at r_cut + s, rebuilt when an atom has moved
s / 2 (the kernels apply the model's cutoff to every edge, so the longer list gives the same energies).  Units: eV, A, fs, amu.
The model has random weights (reference initialisers): the potential is smooth but arbitrary and has NO repulsive core -- after
~150 fs at 300 K some atom pair of the 10^5 collapses (max |F| 3.6 -> 480 eV/A within 40 fs) and no time step integrates that; the
default run therefore stops at 100 + 10 fs, where the total energy error is the O(dt^2) of velocity Verlet: it falls 4x per halving
of dt (profiles/r05_v21_md_loop_*.json: -9.9 / -2.5 / -0.64 eV of 22 600 eV kinetic at 100 fs for dt = 1 / 0.5 / 0.25 fs)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from allegro_amd.nn import HipAllegroModel, neighbor_list  # noqa: E402

KB = 8.617333262e-5        # eV / K
ACC = 9.64853321e-3        # (eV / A / amu) in A / fs^2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4", choices=["c2", "c3", "c4"])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10, help="untimed MD steps before the timed ones (allocator, clocks)")
    ap.add_argument("--dt", type=float, default=1.0)
    ap.add_argument("--temperature", type=float, default=300.0)
    ap.add_argument("--skin", type=float, default=0.0)
    ap.add_argument("--mass", type=float, default=28.0855)
    ap.add_argument("--cross-check", type=int, default=0, metavar="K",
                    help="every K steps evaluate the same (positions, list) through the STAGED pipeline as well and record the difference "
                         "of total energy and forces (diagnostic: which pipeline the default path took is decided per list by max_degree)")
    ap.add_argument("--force-scale", type=float, default=0.0,
                    help="0 = automatic: the random-weight model is scaled so that its rms force is 1 eV/A (a stiffness comparable to a real "
                         "potential at 300 K); the scale multiplies energies and forces alike")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload(args.workload)
    model = HipAllegroModel(**cfg).to(dev)
    staged = None
    if args.cross_check:
        os.environ["AA_FUSED"] = "0"  # (read when a model's plan is created)
        staged = HipAllegroModel(**cfg).to(dev)
        staged.load_state_dict(model.state_dict())
        staged.energy_forces(torch.zeros(2, 3, device=dev), staged.prepare_graph(torch.tensor([[0, 1], [1, 0]], device=dev), torch.zeros(2, dtype=torch.long, device=dev), 2,
                                                                                   torch.zeros(2, 3, device=dev)))
        os.environ.pop("AA_FUSED")
    r_cut = float(cfg["r_max"])
    N = g.num_atoms
    cell = torch.tensor(g.cell, dtype=torch.float64)
    pos = torch.tensor(g.pos, dtype=torch.float32, device=dev)
    types = torch.tensor(g.types, device=dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    vel = torch.randn(N, 3, device=dev, generator=gen) * (KB * args.temperature / args.mass * ACC) ** 0.5  # A / fs
    vel -= vel.mean(0, keepdim=True)

    state = {"graph": None, "pos_ref": None, "rebuilds": 0, "edges": 0, "max_degree": 0}

    def graph_for(p):
        if state["graph"] is None or args.skin == 0.0 or float((p - state["pos_ref"]).square().sum(1).max()) > (0.5 * args.skin) ** 2:
            nl = neighbor_list(p, cell, True, r_cut + args.skin)
            state["graph"] = nl.prepare(types)
            state["pos_ref"] = p.clone()
            state["rebuilds"] += 1
            state["edges"] = nl.num_edges
            state["max_degree"] = max(state["max_degree"], state["graph"].max_degree)
        return state["graph"]

    def forces_of(p):
        e, f = model.energy_forces(p, graph_for(p))
        return e, f

    e_atom, f = forces_of(pos)
    scale = args.force_scale or 1.0 / float(f.square().sum(1).mean().sqrt())
    e_pot0 = float(e_atom.double().sum()) * scale
    f = f * scale

    def kinetic(v):
        return 0.5 * args.mass / ACC * float(v.double().square().sum())

    e0 = e_pot0 + kinetic(vel)
    trace, cross = [], []
    half = 0.5 * args.dt * ACC / args.mass
    t0 = t_prev = 0.0
    step_ms = []
    for step in range(-args.warmup, args.steps):
        if step == 0:
            torch.cuda.synchronize()
            t0 = t_prev = time.perf_counter()
        vel = vel + half * f
        pos = pos + args.dt * vel
        e_atom, f = forces_of(pos)
        f = f * scale
        vel = vel + half * f
        if step >= 0:
            torch.cuda.synchronize()  # (per-step wall times: the median is the robust figure, the slowest steps are listed)
            t_now = time.perf_counter()
            step_ms.append((t_now - t_prev) * 1e3)
            t_prev = t_now
        if step >= 0 and ((step + 1) % max(1, args.steps // 20) == 0 or step + 1 == args.steps):
            trace.append((step + 1, float(e_atom.double().sum()) * scale, kinetic(vel), float(f.square().sum(1).max().sqrt()),
                          float(vel.square().sum(1).max().sqrt())))
        if staged is not None and step >= 0 and (step + 1) % args.cross_check == 0:
            e2, f2 = staged.energy_forces(pos, state["graph"])
            de = (e2.double() - e_atom.double())
            cross.append(dict(step=step + 1, max_degree=state["graph"].max_degree, edges=state["graph"].num_edges,
                              dE_total=float(de.sum()) * scale, max_dE_atom=float(de.abs().max()) * scale, worst_atom=int(de.abs().argmax()),
                              max_dF=float((f2 * scale - f).abs().max())))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    model.check()
    ke = [t[2] for t in trace]
    et = [t[1] + t[2] for t in trace]
    drift = max(abs(x - e0) for x in et)
    ke_mean = sum(ke) / len(ke)
    out = dict(workload=f"{args.workload}: {bench.WORKLOADS[args.workload]['desc']}", atoms=N, edges_last_list=state["edges"], steps=args.steps,
               dt_fs=args.dt, temperature_K=args.temperature, skin_A=args.skin, list_rebuilds=state["rebuilds"], max_degree_seen=state["max_degree"],
               force_scale=scale, ms_per_md_step=wall / args.steps * 1e3, ns_per_day=args.dt * 1e-6 * args.steps / wall * 86400.0,
               ms_per_md_step_median=sorted(step_ms)[len(step_ms) // 2], ns_per_day_at_median=args.dt * 1e-6 / (sorted(step_ms)[len(step_ms) // 2] * 1e-3) * 86400.0,
               slowest_steps=sorted(((round(t, 2), i + 1) for i, t in enumerate(step_ms)), reverse=True)[:6],
               includes="device neighbour list + graph preparation (every rebuild), hot path, velocity-Verlet update, periodic energy read-back",
               e_total_start_eV=e0, max_abs_drift_eV=drift, mean_kinetic_eV=ke_mean, drift_over_mean_kinetic=drift / max(ke_mean, 1e-30),
               drift_per_atom_eV=drift / N,
               cross_check_vs_staged=cross or None,
               trace=[dict(step=s, e_pot=p, e_kin=k, e_tot=p + k, max_force=mf, max_speed=mv) for s, p, k, mf, mv in trace])
    print(json.dumps(out), flush=True)
    # energy conservation: the total energy may fluctuate by O(dt^2) of the kinetic energy, not drift by a sizeable fraction of it
    if not drift / max(ke_mean, 1e-30) < 0.02:
        print(f"md_loop: total energy moved by {drift / ke_mean:.3%} of the mean kinetic energy", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
