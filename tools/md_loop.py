"""NVE molecular dynamics on one MI355X with everything on the device: the metric's "ns/day for a 10^5-atom box" measured as a LOOP --
neighbour list (`aa_nl_*`), graph preparation, the hot path (`aa_model_energy_forces`), velocity-Verlet update -- instead of
0.0864 / t_step, and the energy conservation of that trajectory: an end-to-end check at full size that the forces the hot path
returns are the gradient of the energies it returns (a wrong sign, a missing image, a dropped neighbour contribution or a stale
list shows up as a drift of E_pot + E_kin, not as a number in a table).  The loop itself is `bench.md_loop` (the default bench line
carries a 40-step run of it as `config.md_loop`).

    python tools/md_loop.py [--workload c4] [--steps 100] [--dt 1.0] [--temperature 300] [--skin 0.0 | 0.4] > profiles/rNN_md_loop_c4.json

`--skin 0` (default): the list is rebuilt EVERY step at r_cut (the device list + graph preparation cost ~1 ms at C4, less than
the ~33 % more edges a 0.5 A skin would make every step compute).  `--skin s`: list at r_cut + s, rebuilt when an atom has moved
s / 2 (the kernels apply the model's cutoff to every edge, so the longer list gives the same energies).  Units: eV, A, fs, amu.
The model has random weights (reference initialisers): the potential is smooth but arbitrary and has NO repulsive core -- after
~150 fs at 300 K some atom pair of the 10^5 collapses (max |F| 3.6 -> 480 eV/A within 40 fs) and no time step integrates that; the
default run therefore stops at 100 + 10 fs, where the total energy error is the O(dt^2) of velocity Verlet: it falls 4x per halving
of dt (profiles/r05_v48_md_loop_c4*.json: 12.2 -> 3.07 eV of 21 900 eV kinetic for dt = 1 -> 0.5 fs)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from allegro_amd.nn import HipAllegroModel  # noqa: E402


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
                         "of total energy and forces (which pipeline the default path takes is decided per list)")
    ap.add_argument("--force-scale", type=float, default=0.0,
                    help="0 = automatic: the random-weight model is scaled so that its rms force is 1 eV/A; the scale multiplies energies "
                         "and forces alike")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload(args.workload)
    model = HipAllegroModel(**cfg).to(dev)
    staged = None
    if args.cross_check:
        os.environ["AA_FUSED"] = "0"  # (read when a model's plan is created)
        staged = HipAllegroModel(**cfg).to(dev)
        staged.load_state_dict(model.state_dict())
        staged._select_device(dev)
        staged._ensure_plan()
        os.environ.pop("AA_FUSED")
    out = bench.md_loop(model, torch.tensor(g.pos, dtype=torch.float32, device=dev), torch.tensor(g.types, device=dev),
                        torch.tensor(g.cell, dtype=torch.float64), float(cfg["r_max"]), steps=args.steps, warmup=args.warmup, dt=args.dt,
                        temperature=args.temperature, skin=args.skin, mass=args.mass, force_scale=args.force_scale, cross_check=staged,
                        cross_every=args.cross_check)
    out["workload"] = f"{args.workload}: {bench.WORKLOADS[args.workload]['desc']}"
    print(json.dumps(out), flush=True)
    # energy conservation: the total energy may fluctuate by O(dt^2) of the kinetic energy, not drift by a sizeable fraction of it
    if not out["drift_over_mean_kinetic"] < 0.02:
        print(f"md_loop: total energy moved by {out['drift_over_mean_kinetic']:.3%} of the mean kinetic energy", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
