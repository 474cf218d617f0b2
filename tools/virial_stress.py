"""Stress hunt for the one intermittent of round 2: the native op's virial / forces on the
fused-forward-eligible fixtures, against the ORACLE's strain derivative and the reference's golden forces, many times,
under conditions that make a read-before-write or a missing ordering deterministic:

  * the caching allocator is POISONED before every call (a NaN-filled block of the workspace's size is allocated and
    freed, so the op's fresh `at::empty` workspace starts as NaN instead of as the previous, identical step's values);
  * the Python model runs with aa_plan_options.poison_workspace (AA_POISON=1: the library NaN-fills the workspace before
    every step) through the default (fused) and the staged forward;
  * calls alternate between two HIP streams with unrelated work in flight on the other one;
  * run it again under AMD_SERIALIZE_KERNEL=3 (`AMD_SERIALIZE_KERNEL=3 python tools/virial_stress.py 100`) to separate ordering from data problems.

    python tools/virial_stress.py [iterations]      ->  one line per fixture, exit code 1 on any mismatch"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AA_POISON"] = "1"
os.environ.pop("AA_FUSED", None)

from allegro_amd.export import ExportableAllegro  # noqa: E402
from oracle import restatement as R  # noqa: E402  (checker only)
from tests.golden_utils import load_model_fixture  # noqa: E402
from tests.hip_utils import fixture_data, model_from_fixture  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    noise = torch.randn(2048, 2048, device=dev)
    bad = 0
    for name in ("c2", "c1_L2", "c2_spline"):
        fx = load_model_fixture(name, torch.float32)
        data, sv = fixture_data(fx, torch.float32, dev)
        cfg64 = dict(fx["cfg"], model_dtype="float64")
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in fx["sd"].items()}
        wref = R.allegro_virial(cfg64, sd64, fx["pos"].double(), fx["edge_index"], fx["types"],
                                None if fx["shift_vec"] is None else fx["shift_vec"].double())
        wscale = max(1.0, float(wref.abs().max()))
        fref, eref = fx["out"]["forces"], fx["out"]["atomic_energy"].reshape(-1)
        models = {}
        for mode in ("auto", "staged"):
            if mode == "staged":
                os.environ["AA_FUSED"] = "0"
            else:
                os.environ.pop("AA_FUSED", None)
            models[mode] = model_from_fixture(fx, torch.float32, device=dev)
            models[mode]._ensure_plan()
        os.environ.pop("AA_FUSED", None)
        ex = ExportableAllegro(models["auto"], dev)
        E = data["edge_index"].shape[1]
        perm = torch.randperm(E, generator=torch.Generator().manual_seed(1)).to(dev)
        variants = ((data["edge_index"], sv), (data["edge_index"][:, perm].contiguous(), None if sv is None else sv[perm].contiguous()))
        graphs = {mode: [models[mode].prepare_graph(ei, data["atom_types"], data["pos"].shape[0], s) for ei, s in variants]
                  for mode in models}
        worst = {"op": [0.0, 0.0, 0.0], "auto": [0.0, 0.0, 0.0], "staged": [0.0, 0.0, 0.0]}
        for it in range(iters):
            st = streams[it & 1]
            with torch.cuda.stream(streams[(it + 1) & 1]):
                noise = (noise @ noise).tanh_()  # unrelated work on the other stream
            with torch.cuda.stream(st):
                for vi, (ei, s) in enumerate(variants):
                    # poison the allocator: the op's workspace (at::empty) re-uses this block
                    p = torch.full((8 << 20,), float("nan"), device=dev)
                    del p
                    if it % 3 == 0:  # a fresh list object every third iteration: misses the op's graph cache
                        ei = ei.clone()
                    e_atom, _e_tot, f, vir = ex(data["pos"], ei, data["atom_types"], s)
                    res = {"op": (e_atom.reshape(-1), f, -vir[0])}
                    for mode, m in models.items():
                        e2, f2 = m.energy_forces(data["pos"], graphs[mode][vi])
                        res[mode] = (e2, f2, m.virial(graphs[mode][vi]))
                    st.synchronize()
                    for who, (e_, f_, w_) in res.items():
                        dv = (w_.double().cpu() - wref).abs().max().item() / wscale
                        df = (f_.cpu() - fref).abs().max().item()
                        de = (e_.cpu() - eref).abs().max().item()
                        nonfinite = not (torch.isfinite(w_).all() and torch.isfinite(f_).all() and torch.isfinite(e_).all())
                        worst[who] = [max(worst[who][0], dv), max(worst[who][1], df), max(worst[who][2], de)]
                        if nonfinite or dv > 5e-5 or df > 5e-5 or de > 5e-5:
                            bad += 1
                            print(f"MISMATCH {name} {who} it={it} permuted={vi == 1}: dW(rel)={dv:.3e} dF={df:.3e} dE={de:.3e} "
                                  f"finite={not nonfinite}", flush=True)
        print(name, f"{iters} iterations x 2 edge orders:",
              "; ".join(f"{who}: worst rel dW {w[0]:.2e} dF {w[1]:.2e} dE {w[2]:.2e}" for who, w in worst.items()), flush=True)
    print("virial_stress:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
