"""Operator seam on the layer signatures of standard Allegro stacks: the specialised dense-operand kernels
(allegro_amd/csrc/aa_tp_dense.hip: compile-time Clebsch-Gordan code, lane = channel, one wave per atom and 64-channel
slice) behind `aa_tp_forward` / `aa_tp_backward`, against the table-driven general kernels (AA_TP_GENERIC=1) and the
oracle's eager contraction -- sorted and unsorted scatter indices (tests/nn/test_contract_kernels.py:95-97), 64 / 128 / 256
channels, both dtypes, fp32 1e-5-class / fp64 1e-10 (the reference's kernel-test tolerances, :117,120-134)."""
import pytest
import torch


def _standard_layer_case(l_max, L, layer, mul, dtype, lib, dev, sorted_idxs, monkeypatch, coupling=True):
    """`Contracter.forward` + input gradients on the irreps of a standard Allegro layer (allegro/nn/_allegro.py:101-160):
    the specialised dense-operand kernels (aa_tp_dense.hip) against the table-driven general kernels (AA_TP_GENERIC=1)
    and against the oracle's eager contraction."""
    from allegro_amd import o3
    from allegro_amd.nn import HipContracter, allegro_layer_irreps
    from oracle import restatement as R

    irreps = allegro_layer_irreps(l_max, True, L)
    env = o3.Irreps.spherical_harmonics(l_max, p=-1)
    E, N = 37, 5
    g = torch.Generator().manual_seed(17 + layer)
    idxs = torch.randint(0, N, (E,), generator=g)
    if sorted_idxs:
        idxs = torch.sort(idxs)[0]
    res = {}
    for generic in ("0", "1"):
        monkeypatch.setenv("AA_TP_GENERIC", generic)
        torch.manual_seed(5)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            c = HipContracter(str(irreps[layer]), str(env), str(irreps[layer + 1]), mul=mul, path_channel_coupling=coupling, scatter_factor=0.37)
        finally:
            torch.set_default_dtype(prev)
        c = c.to(dev).eval()
        if lib is not None:
            c._bind_library(lib)
        x1 = torch.randn(E, mul, irreps[layer].dim, dtype=dtype, generator=torch.Generator().manual_seed(21)).to(dev).requires_grad_(True)
        x2 = torch.randn(E, mul, env.dim, dtype=dtype, generator=torch.Generator().manual_seed(22)).to(dev).requires_grad_(True)
        y = c(x1, x2, idxs.to(dev), N)
        gy = torch.randn(y.shape, dtype=dtype, generator=torch.Generator().manual_seed(3)).to(dev)
        g1, g2 = torch.autograd.grad(y, [x1, x2], gy)
        # training mode: the differentiable segmented contraction incl. the path-weight gradient (tp_dense_wgrad_kernel)
        c.train()
        yt = c(x1, x2, idxs.to(dev), N)
        gw = torch.autograd.grad(yt, [c.weights], gy)[0]
        c.eval()
        res[generic] = (y.detach().cpu(), g1.cpu(), g2.cpu(), gw.cpu())
        assert c._get_lib().lib.aa_tp_plan_is_specialised(c._plan(dtype, dev)) == (generic == "0")
        if generic == "0":
            xr1, xr2 = x1.detach().cpu().requires_grad_(True), x2.detach().cpu().requires_grad_(True)
            wr = c.weights.detach().cpu().requires_grad_(True)
            yr = R.contracter_forward(xr1, xr2, idxs, N, wr, c.w3j.cpu(), coupling, 0.37)
            gr1, gr2, grw = torch.autograd.grad(yr, [xr1, xr2, wr], gy.cpu())
            ref = (yr.detach(), gr1, gr2, grw)
    tol = 1e-10 if dtype == torch.float64 else 2e-5
    for what, got, want, gen in zip(("out", "grad x1", "grad x2", "grad weights"), res["0"], ref, res["1"]):
        scale = max(1.0, float(want.abs().max()))
        assert (got - want).abs().max().item() <= tol * scale, f"{what}: specialised kernels vs oracle"
        assert (got - gen).abs().max().item() <= tol * scale, f"{what}: specialised vs general kernels"


@pytest.mark.parametrize("l_max,L,layer,mul,dtype,sorted_idxs", [
    (2, 2, 0, 64, torch.float32, True), (2, 2, 1, 64, torch.float64, False), (1, 2, 0, 128, torch.float64, False),
    (2, 3, 1, 64, torch.float32, False)])
def test_standard_layers_run_the_dense_specialised_kernels_emulated(l_max, L, layer, mul, dtype, sorted_idxs, monkeypatch):
    from tests.hip_utils import emu_lib

    _standard_layer_case(l_max, L, layer, mul, dtype, emu_lib(), torch.device("cpu"), sorted_idxs, monkeypatch)


# path_channel_coupling=False ("p" mode, weights [p] shared by all channels: _contract.py:172-177,244-249; the reference's kernel
# test runs both modes, tests/nn/test_contract_kernels.py:37-40) through the specialised kernels' uncoupled branch
@pytest.mark.parametrize("l_max,L,layer,mul,dtype,sorted_idxs", [
    (2, 2, 0, 64, torch.float64, False), (2, 2, 1, 64, torch.float32, True), (1, 2, 0, 128, torch.float32, False)])
def test_standard_layers_uncoupled_path_weights_emulated(l_max, L, layer, mul, dtype, sorted_idxs, monkeypatch):
    from tests.hip_utils import emu_lib

    _standard_layer_case(l_max, L, layer, mul, dtype, emu_lib(), torch.device("cpu"), sorted_idxs, monkeypatch, coupling=False)


@pytest.mark.gpu
@pytest.mark.parametrize("l_max,L,layer,mul,dtype,sorted_idxs", [
    (2, 2, 0, 64, torch.float32, True), (2, 2, 1, 64, torch.float64, False), (1, 2, 0, 128, torch.float64, False),
    (2, 3, 1, 64, torch.float32, False), (3, 3, 1, 64, torch.float32, True), (1, 2, 1, 256, torch.float32, False)])
def test_standard_layers_uncoupled_path_weights_on_gpu(l_max, L, layer, mul, dtype, sorted_idxs, monkeypatch):
    _standard_layer_case(l_max, L, layer, mul, dtype, None, torch.device("cuda:0"), sorted_idxs, monkeypatch, coupling=False)


@pytest.mark.gpu
@pytest.mark.parametrize("l_max,L,layer,mul,dtype,sorted_idxs", [
    (2, 2, 0, 64, torch.float32, True), (2, 2, 1, 64, torch.float64, False), (1, 2, 0, 128, torch.float64, False),
    (1, 2, 1, 256, torch.float32, False), (2, 3, 0, 64, torch.float64, True), (2, 3, 1, 64, torch.float32, False),
    (2, 3, 2, 128, torch.float32, True), (3, 2, 0, 64, torch.float32, False), (3, 3, 1, 64, torch.float32, True)])
def test_standard_layers_run_the_dense_specialised_kernels_on_gpu(l_max, L, layer, mul, dtype, sorted_idxs, monkeypatch):
    _standard_layer_case(l_max, L, layer, mul, dtype, None, torch.device("cuda:0"), sorted_idxs, monkeypatch)


def _single_gradient_case(lib, dev, dtype):
    """`aa_tp_backward` with one output NULL returns the gradient the full call returns, and `aa_tp_segment_sum`
    is the scale + scatter-sum of _contract.py:195-204 -- on the specialised (mul 64) and the general (mul 8) kernels."""
    from allegro_amd import o3, ops  # noqa: F401  (registers the ops)
    from allegro_amd.nn import HipContracter, allegro_layer_irreps, segments_from_index

    irreps = allegro_layer_irreps(2, True, 2)
    env = o3.Irreps.spherical_harmonics(2, p=-1)
    E, N = 41, 6
    idxs = torch.randint(0, N, (E,), generator=torch.Generator().manual_seed(4))
    for mul, special in ((64, True), (8, False)):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            c = HipContracter(str(irreps[0]), str(env), str(irreps[1]), mul=mul, path_channel_coupling=True, scatter_factor=0.5)
        finally:
            torch.set_default_dtype(prev)
        c = c.to(dev).eval()
        if lib is not None:
            c._bind_library(lib)
        plan, lib_id = c._plan(dtype, dev), c._lib_id
        assert bool(c._get_lib().lib.aa_tp_plan_is_specialised(plan)) == special
        g = torch.Generator().manual_seed(9)
        x1 = torch.randn(E, mul, irreps[0].dim, dtype=dtype, generator=g).to(dev)
        x2 = torch.randn(E, mul, env.dim, dtype=dtype, generator=g).to(dev)
        go = torch.randn(E, mul, irreps[1].dim, dtype=dtype, generator=g).to(dev)
        rowptr, eids = segments_from_index(idxs.to(dev), N)
        x2s = torch.ops.allegro_amd.segment_sum(x2, rowptr, eids, N, 0.5, lib_id)
        want = torch.zeros(N, mul, env.dim, dtype=dtype, device=dev).index_add_(0, idxs.to(dev), x2) * 0.5
        tol = 1e-12 if dtype == torch.float64 else 1e-5
        assert (x2s - want).abs().max().item() <= tol * float(want.abs().max())
        w = c.weights.detach()
        g1, g2 = torch.ops.allegro_amd.tp_backward(go, x1, x2s, w, rowptr, eids, N, 0.5, plan, lib_id)
        o1 = torch.ops.allegro_amd.tp_backward_x1(go, x2s, w, rowptr, eids, N, 0.5, plan, lib_id, irreps[0].dim)
        o2 = torch.ops.allegro_amd.tp_backward_x2(go, x1, w, rowptr, eids, N, 0.5, plan, lib_id, env.dim)
        # (not bitwise: the instantiations contract their FMAs differently, the general kernel sums g2 with LDS atomics)
        for got, ref in ((o1, g1), (o2, g2)):
            assert (got - ref).abs().max().item() <= 10 * tol * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_single_gradients_and_segment_sum_emulated(dtype):
    from tests.hip_utils import emu_lib

    _single_gradient_case(emu_lib(), torch.device("cpu"), dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_single_gradients_and_segment_sum_on_gpu(dtype):
    _single_gradient_case(None, torch.device("cuda:0"), dtype)
