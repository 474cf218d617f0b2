"""Tensor-product kernels that recompute the first-layer x1 weights on the matrix cores (allegro_amd/csrc/aa_tp_mfma.hip):
w0 = EDGE_EMBEDDING @ Wg is rebuilt per 32-edge tile from LDS-resident bf16x3 fragments instead of being re-read from
HBM by every tensor-product kernel.

CPU: the unmodified kernel source in the test-only emulation build against the fp64 oracle on a graph whose segments
span several tiles (degree > 32, partial tiles, an atom without edges), with the launch list checked.
GPU: the same on hardware, the reference's golden vectors, and an A/B against the kernels that read w0 (AA_TP_MFMA=0)."""
import numpy as np
import pytest
import torch

from allegro_amd import graph as G
from allegro_amd.nn import HipAllegroModel
from tests.golden_utils import load_model_fixture
from tests.hip_utils import emu_lib, fixture_data, model_from_fixture
from tests.test_fused import _cfg, _ragged


def _on(monkeypatch, on=True):
    """AA_TP_MFMA is read when the plan is created (the first step of a model)."""
    monkeypatch.setenv("AA_TP_MFMA", "1" if on else "0")


def _dense_cluster(n=40, seed=9):
    """40 atoms at 0.7 spacing: every center atom has more than 32 neighbors inside 3.4 (several MFMA tiles per atom)."""
    rng = np.random.default_rng(seed)
    grid = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(3), indexing="ij"), -1).reshape(-1, 3)[:n]
    pos = grid * 0.7 + rng.uniform(-0.05, 0.05, size=(n, 3)) + 20.0
    pos = np.concatenate([pos, [[50.0, 50.0, 50.0]]])  # isolated: no edges
    cell = np.eye(3) * 60.0
    ei, shift = G.neighbor_list_pbc(pos, cell, 3.4)
    keep = ei[0] < 14  # (14 center atoms keep their segments, the rest are neighbors only: emulation time)
    return pos, cell, ei[:, keep], shift[keep], rng.integers(0, 2, size=n + 1)


def _vs_oracle64(cfg, pos, cell, ei, shift, types, lib, dev):
    """HIP fp32 may not be further from the fp64 oracle on the same (upcast) weights than the fp32 CPU oracle is
    (x2 + a small floor) -- the criterion of tests/test_hip_model.py for fp32 sums."""
    from oracle import restatement as R

    n = pos.shape[0]
    m = HipAllegroModel(**cfg).to(dev)
    if lib is not None:
        m._bind_library(lib)
    sv = torch.tensor(shift @ cell, dtype=torch.float32)
    tt = torch.tensor(types)
    g = m.prepare_graph(torch.tensor(ei).to(dev), tt.to(dev), n, sv.to(dev))
    e, f = m.energy_forces(torch.tensor(pos, dtype=torch.float32, device=dev), g)
    e, f = e.cpu(), f.cpu()
    sd = {k[len("func."):]: v.detach().cpu() for k, v in m.state_dict().items()}
    ref32 = R.allegro_energy_forces(cfg, sd, torch.tensor(pos, dtype=torch.float32), torch.tensor(ei), tt, sv)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = R.allegro_energy_forces(dict(cfg, model_dtype="float64"), sd64, torch.tensor(pos), torch.tensor(ei), tt, sv.double())
    for got, w32, w64 in ((e, ref32["atomic_energy"].reshape(-1), ref64["atomic_energy"].reshape(-1)),
                          (f, ref32["forces"], ref64["forces"])):
        assert torch.isfinite(got).all()
        scale = max(1.0, float(w64.abs().max()))
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu32 = (w32.double() - w64).abs().max().item()
        assert err_hip <= 2.0 * err_cpu32 + 1e-5 * scale, (err_hip, err_cpu32, scale)
    return m



def _assert_launched(m, pos, cell, ei, shift, types, present, absent):
    """The launch list of a step names the kernels that ran."""
    import bench

    g = m.prepare_graph(torch.tensor(ei), torch.tensor(types), pos.shape[0], torch.tensor(shift @ cell, dtype=torch.float32))
    names = [s[0] for s in bench.profile_stages(m, torch.tensor(pos, dtype=torch.float32), g, reps=1)]
    assert all(n in names for n in present) and not any(n in names for n in absent), names


def test_segments_longer_than_one_tile_vs_fp64_oracle_emulated(monkeypatch):
    _on(monkeypatch)
    pos, cell, ei, shift, types = _dense_cluster()
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    assert deg.max() > 32 and deg.min() == 0 and (deg % 32 != 0).any()
    m = _vs_oracle64(_cfg("bessel", True, avg=float(deg.mean()), scale_shift=False), pos, cell, ei, shift, types, emu_lib(),
                     torch.device("cpu"))
    _assert_launched(m, pos, cell, ei, shift, types, ("tp_mfma_fwd_first", "tp_mfma_fwd_last"), ("tp_mom_fwd_first",))


@pytest.mark.gpu
@pytest.mark.parametrize("embed,coupling,l_max", [("bessel", True, 2), ("spline", False, 2), ("bessel", True, 1)])
def test_ragged_and_multi_tile_graphs_vs_fp64_oracle_on_gpu(embed, coupling, l_max, monkeypatch):
    _on(monkeypatch)
    dev = torch.device("cuda:0")
    pos, cell, ei, shift, types = _ragged(dims=(9, 9, 8), keep=0.93, seed=8)
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    _vs_oracle64(_cfg(embed, coupling, l_max=l_max, avg=float(deg.mean())), pos, cell, ei, shift, types, None, dev)
    pos, cell, ei, shift, types = _dense_cluster()
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    assert deg.max() > 32
    _vs_oracle64(_cfg(embed, coupling, l_max=l_max, avg=float(deg.mean()), scale_shift=False), pos, cell, ei, shift, types, None, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c2_spline", "c2_l1", "c2_uncoupled"])
def test_agrees_with_the_w0_reading_kernels_on_gpu(name, monkeypatch):
    """A/B on hardware: the same model and graph with w0 recomputed on the matrix cores and with w0 read from HBM, and
    both against the reference's golden vectors."""
    dev = torch.device("cuda:0")
    fx = load_model_fixture(name, torch.float32)
    data, sv = fixture_data(fx, torch.float32, dev)
    out = []
    for on in (True, False):
        _on(monkeypatch, on)
        m = model_from_fixture(fx, torch.float32, device=dev)
        g = m.prepare_graph(data["edge_index"], data["atom_types"], data["pos"].shape[0], sv)
        out.append(m.energy_forces(data["pos"], g))
    (e, f), (e2, f2) = out
    assert (e - e2).abs().max().item() < 5e-6 and (f - f2).abs().max().item() < 2e-5
    for got, want in ((e.cpu(), fx["out"]["atomic_energy"].reshape(-1)), (f.cpu(), fx["out"]["forces"])):
        assert (got - want).abs().max().item() <= 5e-5 * max(1.0, float(want.abs().max()))
