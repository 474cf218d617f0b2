"""Operator seam: `torch.library` registration of the tensor-product op and the `enable_HipContracter`
model modifier.  CPU: the registration contract (schema, fake kernel, autograd wiring) through `torch.library.opcheck`
and -- in the build container, where the reference is mounted -- the modifier applied to the reference's OWN
AllegroModel (its files imported verbatim behind the leaf shim), energies and autograd forces compared with the
unmodified model.  The kernels run in the test-only CPU emulation build of the HIP sources here; on the GPU the same
ops run the gfx950 library (tests/test_hip_contracter.py)."""
import numpy as np
import pytest
import torch

from tests.hip_utils import emu_lib
from allegro_amd.nn import HipContracter, enable_HipContracter, segments_from_index


def _contracter(coupling=True, mul=4):
    torch.manual_seed(3)
    c = HipContracter("0e + 1o + 2e", "0e + 1o + 2e", "0e + 1o + 2e", mul=mul, path_channel_coupling=coupling,
                      scatter_factor=0.25).double()
    c._bind_library(emu_lib())
    return c


def test_library_op_contract_opcheck():
    c = _contracter()
    E, N = 13, 4
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    x2 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    idxs = torch.randint(0, N, (E,), generator=g)
    rowptr, eids = segments_from_index(idxs, N)
    args = (x1, x2, c.weights.detach(), rowptr, eids, N, 0.25, c._plan(torch.float64), c._lib_id, 9, 9)
    torch.library.opcheck(torch.ops.allegro_amd.tp_forward, args,
                          test_utils=("test_schema", "test_autograd_registration", "test_faketensor"))
    out, x2s = torch.ops.allegro_amd.tp_forward(*args)
    gargs = (torch.randn(out.shape, dtype=torch.float64, generator=g), x1.detach(), x2s, c.weights.detach(), rowptr, eids,
             N, 0.25, c._plan(torch.float64), c._lib_id)
    torch.library.opcheck(torch.ops.allegro_amd.tp_backward, gargs, test_utils=("test_schema", "test_faketensor"))


def test_library_op_gradients_match_finite_differences():
    c = _contracter(coupling=False)
    E, N = 7, 3
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    x2 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    idxs = torch.randint(0, N, (E,), generator=g)
    assert torch.autograd.gradcheck(lambda a, b: c(a, b, idxs, N), (x1, x2), eps=1e-6, atol=1e-7, rtol=1e-6)


def test_ops_refuse_cpu_tensors_without_a_gpu_library():
    """The product op has no CPU implementation: with the default (gfx950) library id it must raise on CPU tensors."""
    c = HipContracter("0e + 1o", "0e + 1o", "0e + 1o", mul=2)
    x = torch.randn(3, 2, 4)
    with pytest.raises(Exception):
        c(x, x, torch.tensor([0, 0, 1]), 2)


def test_modifier_on_reference_model_matches_unmodified_reference():
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference sources are only mounted in the build container")
    from oracle import make_golden as MG

    ref_loader.import_reference()
    from allegro.model import AllegroModel
    from allegro.nn._strided import Contracter
    from nequip.data import AtomicDataDict as ADD

    cfg = MG.test_cfg(coupling=True)
    g = MG.molecule_graph(n=16, box=8.5)
    model = AllegroModel(model_dtype="float64", **cfg).eval()
    data = {ADD.POSITIONS_KEY: torch.tensor(g.pos), ADD.EDGE_INDEX_KEY: torch.tensor(g.edge_index),
            ADD.ATOM_TYPE_KEY: torch.tensor(g.types), ADD.CELL_KEY: torch.tensor(g.cell),
            ADD.EDGE_CELL_SHIFT_KEY: torch.tensor(g.cell_shift, dtype=torch.float64)}
    want = {k: v.detach().clone() for k, v in model(dict(data)).items() if k in ("atomic_energy", "forces")}
    n_before = sum(isinstance(m, Contracter) for m in model.modules())
    sd_keys = list(model.state_dict().keys())
    model = enable_HipContracter(model)            # duck-typed target class
    swapped = [m for m in model.modules() if isinstance(m, HipContracter)]
    assert len(swapped) == n_before == cfg["num_layers"]
    assert not any(isinstance(m, Contracter) for m in model.modules())
    assert list(model.state_dict().keys()) == sd_keys          # checkpoints stay loadable (_contract.py:277)
    for m in swapped:
        m._bind_library(emu_lib())
    got = model(dict(data))
    for k in ("atomic_energy", "forces"):
        assert (got[k] - want[k]).abs().max().item() <= 1e-9 * max(1.0, float(want[k].abs().max())), k


def test_op_traces_under_torch_compile_fullgraph():
    """The op is traceable (fake kernel + functional schema + registered autograd): dynamo/AOTAutograd capture the
    forward and the backward in one graph with no graph break -- what `nequip-compile` needs of an accelerated
    contracter (the reference's op: _flashallegro.py:489,533-670).  Backend aot_eager: no code generation involved."""
    c = _contracter()
    E, N = 13, 4
    g = torch.Generator().manual_seed(2)
    x1 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    x2 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    idxs = torch.sort(torch.randint(0, N, (E,), generator=g))[0]
    rowptr, eids = segments_from_index(idxs, N)
    assert eids is None

    def f(a, b):
        return c._op(a, b, rowptr, None, N, 0.25).square().sum()

    y = f(x1, x2)
    want = torch.autograd.grad(y, [x1, x2])
    fc = torch.compile(f, backend="aot_eager", fullgraph=True)
    y2 = fc(x1, x2)
    got = torch.autograd.grad(y2, [x1, x2])
    assert torch.equal(y.detach(), y2.detach())
    assert all(torch.equal(a, b) for a, b in zip(want, got))
