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


def _check_modifier_on_module_model(dtype, tol, dev, lib):
    """`enable_HipContracter` on a module-structured model built from the t_coupled fixture's reference state_dict:
    every Contracter swapped, state_dict keys unchanged (_contract.py:277), energies / autograd forces equal to the
    reference's golden vectors at the reference's model tolerances (tests/model/test_allegro.py:72-74); in training
    mode the swapped model yields the path-weight gradients of the eager one."""
    from tests.golden_utils import load_model_fixture
    from tests.module_model import EagerContracter, ModuleAllegro
    from allegro_amd.nn import HipContracter

    fx = load_model_fixture("t_coupled", dtype)
    model = ModuleAllegro(fx["cfg"], fx["sd"], dtype).to(dev).eval()
    data = {"pos": fx["pos"].to(dev), "edge_index": fx["edge_index"].to(dev), "atom_types": fx["types"].to(dev)}
    sv = fx["shift_vec"].to(dev)
    ref = fx["out"]

    def check(out):
        for k in ("atomic_energy", "forces"):
            want = ref[k]
            assert (out[k].detach().cpu() - want).abs().max().item() <= tol * max(1.0, float(want.abs().max())), k

    check(model(data, sv))  # the stand-in itself reproduces the reference
    keys = list(model.state_dict().keys())
    n_before = sum(isinstance(mm, EagerContracter) for mm in model.modules())
    # training signal of the eager model (what the swapped one must reproduce): an energy + force-matching loss, i.e.
    # second-order derivatives through every Contracter
    def train_loss(m):
        out = m(data, sv)
        return (out["atomic_energy"] ** 2).sum() + (out["forces"] ** 2).sum()

    model.train()
    want_gw = torch.autograd.grad(train_loss(model), [c.weights for c in model.func.allegro.tps])
    model.eval()
    model = enable_HipContracter(model)
    swapped = [mm for mm in model.modules() if isinstance(mm, HipContracter)]
    assert len(swapped) == n_before == fx["cfg"]["num_layers"]
    assert not any(isinstance(mm, EagerContracter) for mm in model.modules())
    assert list(model.state_dict().keys()) == keys
    if lib is not None:
        for mm in swapped:
            mm._bind_library(lib)
    check(model(data, sv))
    model.train()  # training mode: the arbitrarily differentiable contraction, path-weight gradients included
    got_gw = torch.autograd.grad(train_loss(model), [c.weights for c in model.func.allegro.tps])
    for a, b in zip(got_gw, want_gw):
        assert (a - b).abs().max().item() <= 20 * tol * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 5e-5)])
def test_modifier_on_module_model_matches_golden_emulated(dtype, tol):
    _check_modifier_on_module_model(dtype, tol, torch.device("cpu"), emu_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 5e-5)])
def test_modifier_on_module_model_matches_golden_on_gpu(dtype, tol):
    _check_modifier_on_module_model(dtype, tol, torch.device("cuda:0"), None)


def test_nequip_extension_registers_the_modifier_on_the_reference_contracter():
    """With nequip + allegro importable (here: the reference's files behind the leaf shim) the entry-point hook
    attaches `enable_HipContracter` to the reference's Contracter as a model modifier (_contract.py:253-255)."""
    from oracle import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference sources are only mounted in the build container")
    ref_loader.import_reference()
    import allegro_amd._nequip_ext as ext
    from allegro.nn._strided import Contracter
    from allegro_amd.nn import HipContracter

    ext._STATUS = None
    status = ext.register()
    assert "Contracter.enable_HipContracter" in status, status
    torch.manual_seed(0)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        from e3nn import o3 as e3o3  # (the leaf shim)

        ir = e3o3.Irreps("0e + 1o")
        net = torch.nn.Sequential(Contracter(irreps_in1=ir, irreps_in2=ir, irreps_out=ir, mul=2))
    finally:
        torch.set_default_dtype(old)
    net = Contracter.enable_HipContracter(net)
    assert isinstance(net[0], HipContracter)


@pytest.mark.parametrize("coupling", [True, False])
def test_training_mode_is_differentiable_to_second_order(coupling):
    """Training mode (`.train()`): a force-matching style loss -- a function of the first derivatives wrt x1, differentiated
    again wrt the path weights and both inputs -- equals the eager oracle (autograd through the reference's formulation,
    _contract.py:185-251) in value and in all gradients; plus gradcheck / gradgradcheck of the contraction itself."""
    from oracle import restatement as R

    torch.manual_seed(0)
    c = HipContracter("0e + 1o + 2e", "0e + 1o + 2e", "0e + 1o + 2e", mul=3, path_channel_coupling=coupling,
                      scatter_factor=0.3).double()
    c._bind_library(emu_lib())
    c.train()
    E, N = 9, 4
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(E, 3, 9, dtype=torch.float64, generator=g, requires_grad=True)
    x2 = torch.randn(E, 3, 9, dtype=torch.float64, generator=g, requires_grad=True)
    idx = torch.randint(0, N, (E,), generator=g)

    def loss(fwd, w):
        y = fwd(x1, x2, w)
        (g1,) = torch.autograd.grad(y.square().sum(), x1, create_graph=True)  # "forces"
        return (g1 ** 2).sum() + y.sum()

    l_hip = loss(lambda a, b, w: c(a, b, idx, N), c.weights)
    got = torch.autograd.grad(l_hip, [c.weights, x1, x2])
    w_ref = c.weights.detach().clone().requires_grad_(True)
    l_ref = loss(lambda a, b, w: R.contracter_forward(a, b, idx, N, w, c.w3j, coupling, 0.3), w_ref)
    want = torch.autograd.grad(l_ref, [w_ref, x1, x2])
    assert abs(float(l_hip.detach() - l_ref.detach())) < 1e-10 * max(1.0, abs(float(l_ref.detach())))
    for a, b in zip(got, want):
        assert (a - b).abs().max().item() < 1e-9 * max(1.0, float(b.abs().max()))
    xs = (x1[:4].detach().requires_grad_(True), x2[:4].detach().requires_grad_(True))
    assert torch.autograd.gradcheck(lambda a, b: c._contract(a, b), xs, eps=1e-6, atol=1e-7, rtol=1e-6)
    assert torch.autograd.gradgradcheck(lambda a, b: c._contract(a, b), xs, eps=1e-6, atol=1e-6, rtol=1e-5)


def test_double_backward_through_the_inference_op_fails_loudly():
    """Eval mode runs the fused inference op (scale + scatter + gather + contraction in one kernel, first-order autograd
    for forces).  It registers no second-order formula and must say so instead of returning silently wrong gradients;
    second order is what training mode is for (test above)."""
    c = _contracter()
    c.eval()
    g = torch.Generator().manual_seed(4)
    x1 = torch.randn(5, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    x2 = torch.randn(5, 4, 9, dtype=torch.float64, generator=g, requires_grad=True)
    y = c(x1, x2, torch.tensor([0, 0, 1, 1, 1]), 2)
    (g1,) = torch.autograd.grad(y.sum(), x1, create_graph=True)
    with pytest.raises(RuntimeError):
        g1.square().sum().backward()


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


@pytest.mark.parametrize("assume_sorted", [False, True])
def test_contracter_forward_itself_traces_under_export_and_compile(assume_sorted):
    """`HipContracter.forward` -- not just the op behind it -- must be traceable: the segment bookkeeping
    (bincount / cumsum / sort of the scatter index, the address-keyed cache) is one opaque op with a fake kernel under
    tracing (allegro_amd::segments), so torch.export and torch.compile(fullgraph=True) capture the module as the
    reference's accelerated contracters are captured by `nequip-compile` (_flashallegro.py:725-755).  Random
    (unsorted) and sorted scatter indices, as in tests/nn/test_contract_kernels.py:95-97."""
    c = _contracter().eval()
    c.assume_sorted_idxs = assume_sorted
    E, N = 13, 4
    g = torch.Generator().manual_seed(4)
    x1 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g)
    x2 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g)
    idxs = torch.randint(0, N, (E,), generator=g)
    if assume_sorted:
        idxs = torch.sort(idxs)[0]
    want = c(x1, x2, idxs, N)

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = c

        def forward(self, a, b, i):
            return self.c(a, b, i, N)

    ep = torch.export.export(Wrap(), (x1, x2, idxs))
    targets = [str(n.target) for n in ep.graph.nodes if n.op == "call_function"]
    assert any("allegro_amd.segments" in t for t in targets) and any("allegro_amd.tp_forward" in t for t in targets), targets
    assert torch.equal(ep.module()(x1, x2, idxs), want)
    # other index values through the SAME program (nothing about the traced indices was baked in)
    idxs2 = torch.randint(0, N, (E,), generator=g)
    if assume_sorted:
        idxs2 = torch.sort(idxs2)[0]
    assert torch.equal(ep.module()(x1, x2, idxs2), c(x1, x2, idxs2, N))
    fc = torch.compile(Wrap(), backend="aot_eager", fullgraph=True)
    assert torch.equal(fc(x1, x2, idxs), want)
    torch.library.opcheck(torch.ops.allegro_amd.segments, (idxs, N, assume_sorted), test_utils=("test_schema", "test_faketensor"))


def test_training_path_ops_contract_opcheck():
    """The single-gradient and segment-sum ops of the training path (allegro_amd/ops.py): schema and fake kernels agree with the
    real ones (`torch.library.opcheck`), so they trace under `torch.export` / `torch.compile` like the forward op."""
    c = _contracter()
    E, N = 11, 4
    g = torch.Generator().manual_seed(2)
    x1 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g)
    x2 = torch.randn(E, 4, 9, dtype=torch.float64, generator=g)
    go = torch.randn(E, 4, 9, dtype=torch.float64, generator=g)
    idxs = torch.randint(0, N, (E,), generator=g)
    rowptr, eids = segments_from_index(idxs, N)
    plan, lib_id = c._plan(torch.float64), c._lib_id
    x2s = torch.ops.allegro_amd.segment_sum(x2, rowptr, eids, N, 0.25, lib_id)
    torch.library.opcheck(torch.ops.allegro_amd.segment_sum, (x2, rowptr, eids, N, 0.25, lib_id), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.allegro_amd.tp_backward_x1, (go, x2s, c.weights.detach(), rowptr, eids, N, 0.25, plan, lib_id, 9),
                          test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.allegro_amd.tp_backward_x2, (go, x1, c.weights.detach(), rowptr, eids, N, 0.25, plan, lib_id, 9),
                          test_utils=("test_schema", "test_faketensor"))
