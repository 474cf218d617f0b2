"""Whole-step training mode of `HipAllegroModel` (allegro_amd/training.py): energies and forces of the
differentiable evaluation against the REFERENCE-generated golden outputs, and the gradient of a force-matching loss
(forces differentiated again) with respect to EVERY trainable parameter against autograd through the oracle --
what the reference gets by training its eager model (weights are Parameters: allegro/nn/_allegro.py:192-213,
allegro/nn/_strided/_contract.py:172-177).  CPU: the tensor-product kernels run in the emulation build of the HIP
sources; `-m gpu`: the gfx950 library.  fp64 1e-9, fp32 2e-4 of the largest entry."""
import pytest
import torch

from tests.golden_utils import load_model_fixture
from tests.hip_utils import model_from_fixture


def _loss(out, wf):
    return (out["forces"] * wf).square().sum() + 0.3 * out["total_energy"].sum()


def _training_case(name, dtype, lib, dev):
    from oracle import restatement as R

    fx = load_model_fixture(name, dtype)
    m = model_from_fixture(fx, dtype, lib, dev)
    assert not any(p.requires_grad for p in m.parameters())  # a fresh model is the inference pipeline
    m.train()
    N = fx["pos"].shape[0]
    graph = m.prepare_graph(fx["edge_index"].to(dev), fx["types"].to(dev), N, None if fx["shift_vec"] is None else fx["shift_vec"].to(dev))
    out = m._training_evaluator().forward({"pos": fx["pos"].to(dev)}, graph)
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    for key in ("atomic_energy", "forces"):
        want = fx["out"][key].to(dtype)
        got = out[key].detach().cpu().reshape(want.shape)
        assert (got - want).abs().max().item() <= tol * max(1.0, float(want.abs().max())), f"{name}: {key} vs the reference's golden output"
    wf = torch.linspace(0.5, 1.5, 3 * N, dtype=dtype).reshape(N, 3)
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(_loss(out, wf.to(dev)), [p for p in m.parameters() if p.requires_grad])
    # oracle: the same loss through the restatement, autograd to every float entry of the state_dict
    sd = {k: v.clone() for k, v in fx["sd"].items()}
    for k in names:
        sd[k[len("func."):]].requires_grad_(True)
    pos = fx["pos"].clone().requires_grad_(True)
    e_atom = R.allegro_energy(fx["cfg"], sd, pos, fx["edge_index"], fx["types"], fx["shift_vec"])
    (gp,) = torch.autograd.grad(e_atom.sum(), pos, create_graph=True)
    ref = torch.autograd.grad(_loss({"forces": -gp, "total_energy": e_atom.sum()}, wf), [sd[k[len("func."):]] for k in names])
    assert len(names) >= 8
    for k, g, r in zip(names, grads, ref):
        scale = max(1e-6, float(r.abs().max()))
        assert (g.cpu() - r).abs().max().item() <= tol * scale, f"{name}: d loss / d {k}"
    # back to inference: frozen parameters, the hand-written pipeline
    m.eval()
    assert not any(p.requires_grad for p in m.parameters())


def test_cached_segment_sums_change_no_bit_emulated(monkeypatch):
    """The scaled segment sum of a tensor product's second operand is computed once per operand and reused by every later derivative
    (`ops._TriCtx.x2s_cache`): the loss gradients are bit-identical to recomputing it every time, and the segment-sum op runs less often."""
    from allegro_amd import ops
    from tests.hip_utils import emu_lib

    fx = load_model_fixture("t_coupled", torch.float64)
    N = fx["pos"].shape[0]
    wf = torch.linspace(0.5, 1.5, 3 * N, dtype=torch.float64).reshape(N, 3)
    res, calls = [], []
    real = ops._segment_sum
    for off in ("0", "1"):
        monkeypatch.setenv("AA_TP_NO_X2S_CACHE", off)
        count = {"miss": 0}

        def counted(t, b, _count=count):
            hit = (not ops._NO_X2S_CACHE()) and t._key(b) in t.x2s_cache
            _count["miss"] += 0 if hit else 1
            return real(t, b)

        monkeypatch.setattr(ops, "_segment_sum", counted)
        m = model_from_fixture(fx, torch.float64, emu_lib(), torch.device("cpu"))
        m.train()
        graph = m.prepare_graph(fx["edge_index"], fx["types"], N, fx["shift_vec"])
        out = m._training_evaluator().forward({"pos": fx["pos"]}, graph)
        res.append(torch.autograd.grad(_loss(out, wf), [p for p in m.parameters() if p.requires_grad]))
        calls.append(count["miss"])
    assert all(torch.equal(a, b) for a, b in zip(*res))
    assert calls[0] < calls[1], calls


@pytest.mark.parametrize("name,dtype", [("t_coupled", torch.float64), ("t_uncoupled", torch.float64), ("t_spline_peredge", torch.float64),
                                        ("t_acts", torch.float64), ("t_mish", torch.float64), ("t_shared", torch.float32)])  # (t_acts / t_mish: gelu / mish MLPs on the activation family kernel; c5_small, l_max 3 / 3 layers: GPU list below; 2.5 min emulated)
def test_training_mode_gradients_match_oracle_autograd_emulated(name, dtype):
    from tests.hip_utils import emu_lib

    _training_case(name, dtype, emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype", [("t_coupled", torch.float64), ("t_peredge", torch.float32), ("c2", torch.float32), ("c2", torch.float64),
                                        ("c2_spline", torch.float32), ("c2_uncoupled", torch.float64), ("c2_l3", torch.float32),
                                        ("c2_L3", torch.float32), ("c2_u128", torch.float32), ("c5_small", torch.float64), ("c1_L2", torch.float32),
                                        ("t_acts", torch.float64), ("t_mish", torch.float32)])
def test_training_mode_gradients_match_oracle_autograd_on_gpu(name, dtype):
    _training_case(name, dtype, None, torch.device("cuda:0"))


def _chunked_case(lib, dev, name="t_coupled", dtype=torch.float64, max_edges=70):
    """`ChunkedTrainingStep`: the gradient of a force + energy loss accumulated one block of center atoms at a time equals
    `loss.backward()` through the whole frame (and the forces / energy of its first pass are the inference pipeline's)."""
    fx = load_model_fixture(name, dtype)
    m = model_from_fixture(fx, dtype, lib, dev).train()
    N = fx["pos"].shape[0]
    graph = m.prepare_graph(fx["edge_index"].to(dev), fx["types"].to(dev), N, None if fx["shift_vec"] is None else fx["shift_vec"].to(dev))
    pos = fx["pos"].to(dev)
    tgt = torch.linspace(-0.3, 0.4, 3 * N, dtype=dtype).reshape(N, 3).to(dev)

    def loss_fn(f, e):
        return (f - tgt).square().mean() + 0.05 * (e / N - 0.2).square().sum()

    params = [p for p in m.parameters() if p.requires_grad]
    out = m._training_evaluator().forward({"pos": pos}, graph)
    want = torch.autograd.grad(loss_fn(out["forces"], out["total_energy"]), params)
    for p in params:
        p.grad = None
    step = m.chunked_training_step(graph, max_edges)
    assert len(step.chunks) >= 3
    loss, f, e = step.step(pos, loss_fn)
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    assert abs(float(loss) - float(loss_fn(out["forces"], out["total_energy"]).detach())) <= tol * max(1.0, abs(float(loss)))
    assert (f - out["forces"].detach()).abs().max().item() <= tol * max(1.0, float(f.abs().max()))
    for p, w in zip(params, want):
        assert p.grad is not None and (p.grad - w).abs().max().item() <= tol * max(1e-6, float(w.abs().max()))


def _chunked_isolated_atom_case(lib, dev):
    """An atom without edges that ends up in a block of its own (E_i = shift of its type) contributes r_E to dL/dshifts although
    there is no graph to back-propagate through: the block-wise gradient still equals `loss.backward()` of the frame."""
    from allegro_amd.nn import HipAllegroModel

    fx = load_model_fixture("t_coupled", torch.float64)
    cfg = dict(fx["cfg"], model_dtype="float64", per_type_energy_shifts=[0.3] * len(fx["cfg"]["type_names"]),
               per_type_energy_shifts_trainable=True)
    torch.manual_seed(5)
    m = HipAllegroModel(**cfg).to(dev)
    if lib is not None:
        m._bind_library(lib)
    m.train()
    N = fx["pos"].shape[0] + 1  # (the extra atom sits far away: no edges)
    pos = torch.cat([fx["pos"], fx["pos"].max(0, keepdim=True).values + 50.0]).to(dev)
    types = torch.cat([fx["types"], fx["types"][:1]]).to(dev)
    graph = m.prepare_graph(fx["edge_index"].to(dev), types, N, fx["shift_vec"].to(dev))

    def loss_fn(f, e):
        return f.square().mean() + 0.05 * (e / N - 0.2).square().sum()

    params = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    assert any("shifts" in n for n, _ in params)
    out = m._training_evaluator().forward({"pos": pos}, graph)
    want = torch.autograd.grad(loss_fn(out["forces"], out["total_energy"]), [p for _, p in params])
    for _, p in params:
        p.grad = None
    step = m.chunked_training_step(graph, 3)  # (fewer edges than any atom's segment: one block per atom, the isolated one alone)
    assert step.edge_free_types.numel() == 1
    step.step(pos, loss_fn)
    for (n, p), w in zip(params, want):
        assert p.grad is not None and (p.grad - w).abs().max().item() <= 1e-9 * max(1e-6, float(w.abs().max())), n


def test_chunked_training_step_counts_edge_free_blocks_emulated():
    from tests.hip_utils import emu_lib

    _chunked_isolated_atom_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_chunked_training_step_counts_edge_free_blocks_on_gpu():
    _chunked_isolated_atom_case(None, torch.device("cuda:0"))


def test_chunked_training_step_is_exact_emulated():
    from tests.hip_utils import emu_lib

    _chunked_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype,max_edges", [("t_coupled", torch.float64, 70), ("c2", torch.float32, 500)])
def test_chunked_training_step_is_exact_on_gpu(name, dtype, max_edges):
    _chunked_case(None, torch.device("cuda:0"), name, dtype, max_edges)


def _optimizer_case(lib, dev):
    """Two Adam steps in training mode change the parameters in place; the inference pipeline then evaluates the UPDATED
    model (the packed device weights follow `_version`), equal to the training-mode evaluation of the same parameters."""
    fx = load_model_fixture("t_coupled", torch.float64)
    m = model_from_fixture(fx, torch.float64, lib, dev).train()
    N = fx["pos"].shape[0]
    graph = m.prepare_graph(fx["edge_index"].to(dev), fx["types"].to(dev), N, fx["shift_vec"].to(dev))
    pos = fx["pos"].to(dev)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-2)
    target = torch.zeros(N, 3, dtype=torch.float64, device=dev)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m._training_evaluator().forward({"pos": pos}, graph)
        loss = (out["forces"] - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    f_train = m._training_evaluator().forward({"pos": pos}, graph)["forces"].detach()
    m.eval()
    _e, f_eval = m.energy_forces(pos, graph)
    assert (f_eval - f_train).abs().max().item() <= 1e-9 * max(1.0, float(f_train.abs().max()))


def test_optimizer_steps_then_inference_emulated():
    from tests.hip_utils import emu_lib

    _optimizer_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_optimizer_steps_then_inference_on_gpu():
    _optimizer_case(None, torch.device("cuda:0"))


def _public_route_case(lib, dev):
    """`model.train(); model(data)` -- the AtomicDataDict route with a cell: forces equal the golden output, stress equals
    the oracle's strain derivative / volume, and both stay attached to the graph."""
    from oracle import restatement as R

    fx = load_model_fixture("t_peredge", torch.float64)
    m = model_from_fixture(fx, torch.float64, lib, dev).train()
    cell = torch.eye(3, dtype=torch.float64) * 1.0  # shift_vec = edge_cell_shift @ cell with a unit cell
    data = {"pos": fx["pos"].to(dev), "edge_index": fx["edge_index"].to(dev), "atom_types": fx["types"].to(dev),
            "cell": cell.to(dev), "edge_cell_shift": fx["shift_vec"].to(dev)}
    out = m(data)
    want = fx["out"]["forces"].to(torch.float64)
    assert (out["forces"].detach().cpu() - want).abs().max().item() <= 1e-9 * max(1.0, float(want.abs().max()))
    vir = R.allegro_virial(fx["cfg"], fx["sd"], fx["pos"], fx["edge_index"], fx["types"], fx["shift_vec"])
    assert (out["stress"].detach().cpu().reshape(3, 3) - vir).abs().max().item() <= 1e-9 * max(1.0, float(vir.abs().max()))
    assert out["forces"].requires_grad and out["stress"].requires_grad and out["total_energy"].requires_grad
    with torch.no_grad():  # validation inside a training loop: the inference pipeline
        out2 = m(data)
    assert not out2["forces"].requires_grad
    assert (out2["forces"].cpu() - want).abs().max().item() <= 1e-9 * max(1.0, float(want.abs().max()))


def test_training_mode_public_route_with_stress_emulated():
    from tests.hip_utils import emu_lib

    _public_route_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_training_mode_public_route_with_stress_on_gpu():
    _public_route_case(None, torch.device("cuda:0"))
