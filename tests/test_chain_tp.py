"""Forward with the tensor-track scalars evaluated inside the linear-layer chains that produce w0 (plan option chain_tp,
AA_CHAIN_TP=1; gemm_chain_bf16x3_kernel<.., TPX>): the moments kernels only form the per-atom Clebsch-Gordan vectors,
the chain gathers them by center atom behind every irrep's tile pair, scal0 / scal1 never reach HBM.

CPU: the unmodified kernel sources in the test-only emulation build against the fp64 oracle (ragged graph, multi-atom
tiles, an atom without edges) with the launch list checked.  GPU: the same, the reference's golden vectors and an A/B
against the staged forward."""
import numpy as np
import pytest
import torch

from tests.golden_utils import load_model_fixture
from tests.hip_utils import emu_lib, fixture_data, model_from_fixture
from tests.test_fused import _cfg, _ragged
from tests.test_tp_mfma import _assert_launched, _dense_cluster, _vs_oracle64


def _on(monkeypatch, on=True):
    """AA_CHAIN_TP is read when the plan is created (the first step of a model)."""
    monkeypatch.setenv("AA_CHAIN_TP", "1" if on else "0")
    monkeypatch.setenv("AA_FUSED", "0")  # (the fused forward takes precedence over this staged variant)


def test_ragged_graph_vs_fp64_oracle_emulated(monkeypatch):
    _on(monkeypatch)
    pos, cell, ei, shift, types = _ragged()
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    assert deg.min() == 0 and len(set(deg.tolist())) > 6
    m = _vs_oracle64(_cfg("spline", False, avg=float(deg.mean())), pos, cell, ei, shift, types, emu_lib(), torch.device("cpu"))
    _assert_launched(m, pos, cell, ei, shift, types, ("tp_mom_vec_first", "tp_mom_vec_last"), ("tp_mom_fwd_first",))


@pytest.mark.gpu
@pytest.mark.parametrize("embed,coupling,l_max", [("bessel", True, 2), ("spline", False, 2), ("bessel", True, 1)])
def test_ragged_and_dense_graphs_vs_fp64_oracle_on_gpu(embed, coupling, l_max, monkeypatch):
    _on(monkeypatch)
    dev = torch.device("cuda:0")
    pos, cell, ei, shift, types = _ragged(dims=(9, 9, 8), keep=0.93, seed=8)
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    _vs_oracle64(_cfg(embed, coupling, l_max=l_max, avg=float(deg.mean())), pos, cell, ei, shift, types, None, dev)
    pos, cell, ei, shift, types = _dense_cluster()
    deg = np.bincount(ei[0], minlength=pos.shape[0])
    _vs_oracle64(_cfg(embed, coupling, l_max=l_max, avg=float(deg.mean()), scale_shift=False), pos, cell, ei, shift, types, None, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c2_spline", "c2_l1", "c2_uncoupled"])
def test_agrees_with_the_staged_forward_on_gpu(name, monkeypatch):
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
