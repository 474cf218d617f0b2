"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by oracle/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_FIXTURES = ["c1_L1", "c1_L2", "c2", "t_coupled", "t_uncoupled", "t_peredge", "c5_small", "t_spline",
                  "t_spline_peredge", "c2_spline", "c2_l3", "c2_l1", "c2_L3", "c2_u128", "c2_uncoupled", "t_acts", "t_mish", "t_shared",
                  "c2_shared"]


def load_model_fixture(name, dtype=torch.float64):
    z = np.load(os.path.join(GOLDEN_DIR, f"model_{name}.npz"))
    cfg = json.loads(str(z["cfg_json"]))
    sd = {k[3:]: torch.tensor(z[k]).to(dtype) if z[k].dtype.kind == "f" else torch.tensor(z[k])
          for k in z.files if k.startswith("sd/")}
    tag = "out64/" if dtype == torch.float64 else "out32/"
    out = {k[len(tag):]: torch.tensor(z[k]) for k in z.files if k.startswith(tag)}
    fx = dict(cfg=cfg, sd=sd, pos=torch.tensor(z["pos"]).to(dtype), edge_index=torch.tensor(z["edge_index"]),
              types=torch.tensor(z["types"]), out=out,
              shift_vec=torch.tensor(z["shift_vec"]).to(dtype) if "shift_vec" in z.files else None)
    return fx


def load_contract_cases():
    z = np.load(os.path.join(GOLDEN_DIR, "contract_cases.npz"))
    cases = []
    for tag in z["names"]:
        c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(f"{tag}/")}
        c["meta"] = json.loads(str(c["meta"]))
        cases.append(c)
    return cases
