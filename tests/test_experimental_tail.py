"""EXPERIMENTAL build only (AA_BUILD_EXPERIMENTAL=1; skipped otherwise): the fused per-atom-tile reverse tail
(allegro_amd/csrc/aa_fused_bwd.hip) -- layer-0 tensor-product reverse + first-stage /
scalar_embed_mlp reverse + edge reverse in one launch.  Parity-green; measured slower than the staged tail on MI355X, so
it is not part of the product library.  These tests keep the recorded measurements reproducible:

    AA_BUILD_EXPERIMENTAL=1 python -m pytest tests/test_experimental_tail.py            (emulated)
    AA_BUILD_EXPERIMENTAL=1 python -m pytest tests/test_experimental_tail.py -m gpu     (MI355X)"""
import os

import pytest
import torch

from tests.golden_utils import load_model_fixture
from tests.hip_utils import emu_lib, fixture_data, model_from_fixture

pytestmark = pytest.mark.skipif(os.environ.get("AA_BUILD_EXPERIMENTAL", "0")[:1] != "1",
                                reason="the fused reverse tail is only compiled under AA_BUILD_EXPERIMENTAL=1")


def _ab(name, lib, dev, monkeypatch, atoms=None):
    import bench

    fx = load_model_fixture(name, torch.float32)
    data, sv = fixture_data(fx, torch.float32, dev)
    ei, s = data["edge_index"], sv
    if atoms is not None:  # (emulation time: the first few center atoms)
        keep = ei[0] < atoms
        ei, s = ei[:, keep], None if sv is None else sv[keep]
    out = {}
    for mode in ("1", "2", "0"):
        monkeypatch.setenv("AA_FUSED_TAIL", mode)
        m = model_from_fixture(fx, torch.float32, lib, device=dev)
        g = m.prepare_graph(ei, data["atom_types"], data["pos"].shape[0], s)
        e, f = m.energy_forces(data["pos"], g)
        w = m.virial(g)
        names = [st[0] for st in bench.profile_stages(m, data["pos"], g, reps=1)]
        assert ("fused_bwd_tail" in names) == (mode != "0") and ("tp_mom_bwd_first" in names) == (mode == "0"), names
        assert ("edge_backward" in names) == (mode != "1"), names
        out[mode] = (e.cpu(), f.cpu(), w.cpu())
    for mode in ("1", "2"):
        assert (out[mode][1] - out["0"][1]).abs().max().item() < 2e-5 * max(1.0, float(out["0"][1].abs().max()))
        assert (out[mode][2] - out["0"][2]).abs().max().item() < 5e-5 * max(1.0, float(out["0"][2].abs().max()))
    if atoms is None:
        assert (out["1"][1] - fx["out"]["forces"]).abs().max().item() <= 5e-5 * max(1.0, float(fx["out"]["forces"].abs().max()))


@pytest.mark.parametrize("name", ["c2", "c1_L2"])
def test_fused_reverse_tail_matches_the_staged_tail_emulated(name, monkeypatch):
    _ab(name, emu_lib(), torch.device("cpu"), monkeypatch, atoms=9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c1_L2", "c2_spline", "c2_uncoupled", "c2_shared", "c2_l1"])
def test_fused_reverse_tail_matches_the_staged_tail_and_the_golden_on_gpu(name, monkeypatch):
    _ab(name, None, torch.device("cuda:0"), monkeypatch)
