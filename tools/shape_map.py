"""Performance map of the model space: every hyper-parameter set of the 19 golden fixtures on ONE box.

For each fixture's constructor arguments (tests/golden/model_*.npz: the reference's tutorial / test / BASELINE shapes and one
fixture per kernel family) the model is built with random weights in BOTH dtypes on the C3 box (bulk Si 11^3 cells, 10 648 atoms,
~2.98e5 edges, r_cut 5 A; multi-species fixtures get species assigned at random) -- or the C4 box with `--box c4` -- and one step is
timed.  Printed per row: ms/step, which pipeline the plan selected (aa_model_plan_describe + the launch list), the dominant kernel
and its share of the step, the algorithmic flop/edge of SURVEY 8d, and `x_c3` = time relative to the flop-scaled time of the
tuned shape (C2's hyper-parameters on the same box): 1.0 = as efficient as the headline path, 3.0 = a 3x cliff.

    python tools/shape_map.py [--box c3|c4] [--dtype float32|float64|both] [--only name,name] > profiles/rNN_shape_map_c3.md
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from allegro_amd import graph as G  # noqa: E402
from allegro_amd.nn import HipAllegroModel, PreparedGraph  # noqa: E402
from tests.golden_utils import MODEL_FIXTURES, load_model_fixture  # noqa: E402


# Shapes published Allegro configurations use that no golden fixture has (timing only; the kernels they select are covered by the
# fixture families and the emulated ragged / padded stacks): wider and deeper latent MLPs, 3 layers at l_max 3, 128 scalars.
def _si(l_max, L, u, S, Hl, dl, Hr=None, He=None):
    return dict(type_names=["Si"], r_max=5.0, l_max=l_max, parity=True, num_layers=L, num_scalar_features=S, num_tensor_features=u,
                radial_chemical_embed=dict(bench.BESSEL), radial_chemical_embed_dim=S, scalar_embed_mlp_hidden_layers_depth=1,
                scalar_embed_mlp_hidden_layers_width=He or S, allegro_mlp_hidden_layers_depth=dl, allegro_mlp_hidden_layers_width=Hl,
                readout_mlp_hidden_layers_depth=1, readout_mlp_hidden_layers_width=Hr or Hl, avg_num_neighbors=28.0,
                tp_path_channel_coupling=True, seed=456)


EXTRA_SHAPES = {
    "x_lat2x128": _si(2, 2, 64, 64, 128, 2),         # latent MLP 2 x 128 (the depth / width most published configs use)
    "x_lat1x128": _si(2, 2, 64, 64, 128, 1),
    "x_l3_L3_u64": _si(3, 3, 64, 64, 64, 1),         # the C5 depth at 64 features, fp32
    "x_S128_u64": _si(2, 2, 64, 128, 128, 1),
    "x_u32_lat2x128": _si(2, 2, 32, 64, 128, 2),
    "x_L1_u64": _si(2, 1, 64, 64, 64, 1),
}


def pipeline_name(desc, names):
    if desc.get("fused_forward") and any(n.startswith("fused_fwd") for n in names):
        return "fused forward + chains"
    if desc.get("operator_path"):
        return "operator kernels, " + ("slot form" if desc.get("slot_form") else "single layers")
    if desc.get("chain_gemm") and desc.get("moments"):
        return "staged: chains + moments"
    if desc.get("chain_gemm"):
        return "staged: chains + per-edge TP"
    if desc.get("moments"):
        return "single layers + moments"
    return "single layers + per-edge TP"


def run_one(name, cfg, g, dtype, dev, steps):
    cfg = dict(cfg)
    cfg["model_dtype"] = dtype
    cfg["r_max"] = 5.0
    cfg.pop("per_edge_type_cutoff", None)  # (the box has one length scale; per-type cutoffs are a prologue detail)
    cfg["avg_num_neighbors"] = g.num_edges / g.num_atoms
    T = len(cfg["type_names"])
    tdt = torch.float32 if dtype == "float32" else torch.float64
    model = HipAllegroModel(**cfg).to(dev)
    types = np.random.default_rng(3).integers(0, T, size=g.num_atoms)
    sv = g.shift_vec()
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(types, device=dev), g.num_atoms,
                          torch.tensor(sv, dtype=tdt, device=dev))
    pos = torch.tensor(g.pos, dtype=tdt, device=dev)
    for _ in range(3):
        model.energy_forces(pos, graph)
    torch.cuda.synchronize()
    n = steps
    t0 = time.perf_counter()
    for _ in range(n):
        model.energy_forces(pos, graph)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    model.check()
    stages = bench.profile_stages(model, pos, graph, reps=2)
    by = {}
    for nm, ms, _, _ in stages:
        by[nm] = by.get(nm, 0.0) + ms
    dom, dom_ms = max(by.items(), key=lambda kv: kv[1])
    tot = sum(by.values())
    desc = model.describe_plan()
    roof = bench.step_roofline(cfg, g.num_edges, t, stages, dtype)
    rec = dict(name=name, dtype=dtype, ms=t * 1e3, pipeline=pipeline_name(desc, [s[0] for s in stages]), launches=len(stages),
               dominant=dom, dominant_share=dom_ms / max(tot, 1e-12), flop_per_edge=roof["flop_per_edge"],
               frac_fused_roof=roof["frac_of_fused_compute_roof"], l_max=cfg["l_max"], L=cfg["num_layers"],
               u=cfg["num_tensor_features"], S=cfg["num_scalar_features"], T=T,
               mlp=f"{cfg.get('allegro_mlp_hidden_layers_depth', 1)}x{cfg.get('allegro_mlp_hidden_layers_width', 64)}")
    del model, graph, pos
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--box", default="c3", choices=["c3", "c4"])
    ap.add_argument("--dtype", default="both")
    ap.add_argument("--only", default="")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g, _ = bench.make_workload(args.box)
    names = [n for n in list(MODEL_FIXTURES) + list(EXTRA_SHAPES) if not args.only or n in args.only.split(",")]
    dtypes = ["float32", "float64"] if args.dtype == "both" else [args.dtype]
    rows = []
    seen = set()
    for name in names:
        cfg = EXTRA_SHAPES[name] if name in EXTRA_SHAPES else load_model_fixture(name)["cfg"]
        key = json.dumps({k: v for k, v in cfg.items() if k not in ("seed", "per_edge_type_cutoff", "avg_num_neighbors", "r_max")}, sort_keys=True)
        if key in seen:  # (fixtures that differ only in seed / cutoffs / geometry are one shape)
            continue
        seen.add(key)
        for dt in dtypes:
            try:
                rows.append(run_one(name, cfg, g, dt, dev, args.steps))
            except Exception as ex:  # noqa: BLE001  (a shape the box cannot hold, e.g. workspace: report it, keep going)
                rows.append(dict(name=name, dtype=dt, error=repr(ex)[:160]))
            r = rows[-1]
            print(f"# {name} {dt}: " + (f"{r['ms']:.3f} ms  {r['pipeline']}" if "ms" in r else r["error"]), file=sys.stderr, flush=True)
    ref = {r["dtype"]: r for r in rows if r["name"] == "c2" and "ms" in r}
    print(f"| fixture shape | dtype | l_max | L | u / S | species | latent MLP | ms/step ({args.box}) | pipeline | launches | dominant kernel (share) | "
          "kflop/edge | x flop-scaled c2 fp32 |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    base = ref.get("float32")
    for r in rows:
        if "ms" not in r:
            print(f"| {r['name']} | {r['dtype']} | | | | | | failed: {r['error']} | | | | | |")
            continue
        rel = (r["ms"] / base["ms"]) / (r["flop_per_edge"] / base["flop_per_edge"]) if base else float("nan")
        print(f"| {r['name']} | {r['dtype'][5:]} | {r['l_max']} | {r['L']} | {r['u']} / {r['S']} | {r['T']} | {r['mlp']} | {r['ms']:.3f} | {r['pipeline']} | "
              f"{r['launches']} | {r['dominant']} ({100 * r['dominant_share']:.0f} %) | {r['flop_per_edge'] / 1e3:.0f} | {rel:.2f} |")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
