"""Repeat the native-op vs model comparison of tests/test_export.py and print the differences (hunting an intermittent)."""
import sys

import torch

sys.path.insert(0, ".")
import os

os.environ.setdefault("AA_FUSED", "0")  # the Python model: staged (as under tests/conftest.py); the native op: its own defaults
from tests.test_export import _exportable

dev = torch.device("cuda:0")
for name in ("c2", "c1_L2"):
    fx, m, data, sv, ex = _exportable(name, torch.float32, dev)
    ref = fx["out"]
    perm = torch.randperm(data["edge_index"].shape[1], generator=torch.Generator().manual_seed(1)).to(dev)
    worst = [0.0, 0.0, 0.0]
    for it in range(40):
        for ei, s in ((data["edge_index"], sv), (data["edge_index"][:, perm], None if sv is None else sv[perm])):
            e_atom, e_tot, f, vir = ex(data["pos"], ei, data["atom_types"], s)
            g = m.prepare_graph(ei, data["atom_types"], data["pos"].shape[0], s)
            m.energy_forces(data["pos"], g)
            w = m.virial(g)
            dv = (vir[0] + w).abs().max().item() / max(1.0, float(w.abs().max()))
            df = (f.cpu() - ref["forces"]).abs().max().item()
            de = (e_atom.cpu().reshape(-1) - ref["atomic_energy"].reshape(-1)).abs().max().item()
            worst = [max(worst[0], dv), max(worst[1], df), max(worst[2], de)]
            if dv > 5e-5 or df > 5e-5:
                print(f"{name} it={it} perm={ei is not data['edge_index']}: dvirial(rel)={dv:.3e} dF={df:.3e} dE={de:.3e}", flush=True)
    print(name, "worst rel dvirial %.3e dF %.3e dE %.3e" % tuple(worst), flush=True)
