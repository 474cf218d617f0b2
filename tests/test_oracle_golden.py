"""CPU: pins oracle/restatement.py against the golden vectors produced by the reference itself."""
import pytest
import torch

from oracle import restatement as R
from tests.golden_utils import MODEL_FIXTURES, load_model_fixture, load_contract_cases


@pytest.mark.parametrize("name", MODEL_FIXTURES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_restatement_matches_reference_outputs(name, dtype):
    fx = load_model_fixture(name, dtype)
    out = R.allegro_energy_forces(fx["cfg"], fx["sd"], fx["pos"], fx["edge_index"], fx["types"], fx["shift_vec"])
    # the reference's own whole-model tolerances (tests/model/test_allegro.py:72-74), relative to the output scale
    tol = {torch.float64: 1e-10, torch.float32: 5e-5}[dtype]
    for k in ("atomic_energy", "forces"):
        ref = fx["out"][k]
        scale = max(1.0, float(ref.abs().max()))
        assert (out[k] - ref).abs().max().item() <= tol * scale, k
    ref = fx["out"]["total_energy"]
    assert abs(float(out["total_energy"].sum() - ref.sum())) <= tol * max(1.0, float(fx["out"]["atomic_energy"].abs().sum()))


@pytest.mark.parametrize("name", ["c2", "t_coupled"])
def test_chunked_equals_whole(name):
    fx = load_model_fixture(name, torch.float64)
    a = R.allegro_energy_forces(fx["cfg"], fx["sd"], fx["pos"], fx["edge_index"], fx["types"], fx["shift_vec"])
    b = R.allegro_energy_forces_chunked(fx["cfg"], fx["sd"], fx["pos"], fx["edge_index"], fx["types"], fx["shift_vec"],
                                        max_edges=300)
    assert (a["forces"] - b["forces"]).abs().max() < 1e-10
    assert (a["atomic_energy"] - b["atomic_energy"]).abs().max() < 1e-10


def test_contracter_restatement_matches_reference_cases():
    for c in load_contract_cases():
        m = c["meta"]
        x1 = torch.tensor(c["x1"], requires_grad=True)
        x2 = torch.tensor(c["x2"], requires_grad=True)
        y = R.contracter_forward(x1, x2, torch.tensor(c["idxs"]), m["num_atoms"], torch.tensor(c["weights"]),
                                 torch.tensor(c["w3j"]), m["coupling"], m["scatter_factor"])
        g1, g2 = torch.autograd.grad(y, [x1, x2], torch.tensor(c["gout"]))
        assert (y - torch.tensor(c["out"])).abs().max() < 1e-10
        assert (g1 - torch.tensor(c["gx1"])).abs().max() < 1e-10
        assert (g2 - torch.tensor(c["gx2"])).abs().max() < 1e-10
