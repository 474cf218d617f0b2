"""GPU: HipContracter vs the reference's eager Contracter at the shapes and tolerances of the reference's
own kernel test (tests/nn/test_contract_kernels.py:93-134): 17 edges, 5 atoms, random idxs, mul 3/8,
both weight modes, fp32 1e-5 / fp64 1e-10, forward, both input gradients and the path-weight gradient."""
import pytest
import torch

from allegro_amd.nn import HipContracter
from tests.golden_utils import load_contract_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
def test_contracter_forward_and_input_grads(dtype, tol):
    dev = torch.device("cuda:0")
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        for c in load_contract_cases():
            m = c["meta"]
            mod = HipContracter(m["irreps_in1"], m["irreps_in2"], m["irreps_out"], m["mul"],
                                path_channel_coupling=m["coupling"], scatter_factor=m["scatter_factor"])
            mod.load_state_dict({"weights": torch.tensor(c["weights"]).to(dtype), "w3j": torch.tensor(c["w3j"]).to(dtype)})
            mod = mod.to(dev)
            x1 = torch.tensor(c["x1"]).to(dtype).to(dev).requires_grad_(True)
            x2 = torch.tensor(c["x2"]).to(dtype).to(dev).requires_grad_(True)
            y = mod(x1, x2, torch.tensor(c["idxs"]).to(dev), torch.tensor([m["num_atoms"]]))
            g1, g2, gw = torch.autograd.grad(y, [x1, x2, mod.weights], torch.tensor(c["gout"]).to(dtype).to(dev))
            for got, want in ((y, c["out"]), (g1, c["gx1"]), (g2, c["gx2"])):
                assert (got.double().cpu() - torch.tensor(want)).abs().max().item() < tol
            want = torch.tensor(c["gw"])  # path-weight gradient vs the reference's eager autograd
            assert gw.shape == mod.weights.shape
            assert (gw.double().cpu() - want).abs().max().item() < tol * max(1.0, float(want.abs().max()))
    finally:
        torch.set_default_dtype(old)


def test_contract_only_seam_b1():
    """Contracter._contract (seam B1): contraction without the scatter/gather."""
    from oracle import restatement as R

    dev = torch.device("cuda:0")
    c = load_contract_cases()[0]
    m = c["meta"]
    torch.set_default_dtype(torch.float64)
    try:
        mod = HipContracter(m["irreps_in1"], m["irreps_in2"], m["irreps_out"], m["mul"], path_channel_coupling=m["coupling"])
        mod.load_state_dict({"weights": torch.tensor(c["weights"]), "w3j": torch.tensor(c["w3j"])})
        mod = mod.to(dev)
        x1, x2 = torch.tensor(c["x1"]).to(dev), torch.tensor(c["x2"]).to(dev)
        y = mod._contract(x1, x2).cpu()
        want = R.contract(torch.tensor(c["x1"]), torch.tensor(c["x2"]), torch.tensor(c["weights"]), torch.tensor(c["w3j"]), m["coupling"])
        assert (y - want).abs().max() < 1e-10
    finally:
        torch.set_default_dtype(torch.float32)


def test_library_op_opcheck_and_compile_trace_on_gpu():
    """The registered op on the gfx950 library: opcheck (schema / fake / autograd registration) and a fullgraph
    torch.compile(aot_eager) trace of forward+backward that reproduces the eager launches."""
    from allegro_amd.nn import segments_from_index

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    c = HipContracter("0e + 1o + 2e", "0e + 1o + 2e", "0e + 1o + 2e", mul=64, scatter_factor=0.2).to(dev)
    E, N = 300, 11
    x1 = torch.randn(E, 64, 9, device=dev, requires_grad=True)
    x2 = torch.randn(E, 64, 9, device=dev, requires_grad=True)
    idxs = torch.sort(torch.randint(0, N, (E,), device=dev))[0]
    rowptr, eids = segments_from_index(idxs, N)
    args = (x1, x2, c.weights.detach(), rowptr, eids, N, 0.2, c._plan(torch.float32), 0, 9, 9)
    torch.library.opcheck(torch.ops.allegro_amd.tp_forward, args,
                          test_utils=("test_schema", "test_autograd_registration", "test_faketensor"))

    def f(a, b):
        return c._op(a, b, rowptr, None, N, 0.2).square().sum()

    want = torch.autograd.grad(f(x1, x2), [x1, x2])
    got = torch.autograd.grad(torch.compile(f, backend="aot_eager", fullgraph=True)(x1, x2), [x1, x2])
    for a, b in zip(want, got):  # (the stand-alone operator's segment sums use atomics: equal to rounding, not bitwise)
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, float(a.abs().max()))

