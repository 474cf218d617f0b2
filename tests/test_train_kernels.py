"""The training-path kernels of csrc/aa_train.hip against plain PyTorch (the checker): `aa_linear_wgrad` (weight gradient of a
bias-free linear layer, reduced over the edges in fixed-order slabs on the matrix cores) and `aa_weighted_channels`
(MakeWeightedChannels, allegro/nn/_strided/_channels.py:44-63, as a bilinear form with its two partial contractions), incl.
their closure under differentiation: second derivatives through `ops.linear` / `ops.weighted_channels` equal autograd's
through the eager forms.  CPU: the emulation build of the same sources; `-m gpu`: the gfx950 library."""
import pytest
import torch

from allegro_amd import ops


def _lib_id(lib):
    return 0 if lib is None else ops.register_library(lib)


def _wgrad_case(lib, dev):
    lid = _lib_id(lib)
    g = torch.Generator().manual_seed(4)
    for dtype, tol in ((torch.float32, 2e-6), (torch.float64, 1e-13)):
        for E, K, N, pad in ((0, 8, 16, 0), (5, 8, 64, 0), (777, 64, 64, 0), (1030, 192, 64, 16), (2049, 17, 130, 3), (300, 256, 1, 0), (17933, 4, 5, 0)):  # (last: 71 slabs -> both launches of the column sum, ragged last chunk)
            xb = torch.randn(E, K + pad, generator=g, dtype=dtype).to(dev)
            gb = torch.randn(E, N + pad, generator=g, dtype=dtype).to(dev)
            x, gg = xb[:, :K], gb[:, pad:]  # row-strided views: the kernel takes the strides, no copy
            got = torch.ops.allegro_amd.linear_wgrad(x, gg, lid)
            want = x.double().t() @ gg.double()
            scale = max(1.0, float(want.abs().max()))
            assert got.shape == (K, N) and (got.double() - want).abs().max().item() <= tol * scale * max(1, E) ** 0.5, (dtype, E, K, N)
            again = torch.ops.allegro_amd.linear_wgrad(x, gg, lid)
            assert torch.equal(got, again)  # fixed summation order: the same bits every call


def _second_order_case(lib, dev):
    """A force-matching loss differentiates a first derivative again: grad-of-grad through the op pair (_MM, _XtG) and the
    weighted-channel triple equals autograd through matmul / broadcasting."""
    lid = _lib_id(lib)
    dtype = torch.float64
    g = torch.Generator().manual_seed(9)
    E, K, H, N, u, l_max = 37, 8, 24, 12, 6, 2
    D = (l_max + 1) ** 2
    x0 = torch.randn(E, K, generator=g, dtype=dtype).to(dev)
    W1 = torch.randn(K, H, generator=g, dtype=dtype).to(dev)
    W2 = torch.randn(H, u * (l_max + 1), generator=g, dtype=dtype).to(dev)
    sh0 = torch.randn(E, D, generator=g, dtype=dtype).to(dev)
    c = torch.randn(E, u, D, generator=g, dtype=dtype).to(dev)

    def run(hand):
        x = x0.clone().requires_grad_(True)
        sh = sh0.clone().requires_grad_(True)
        w1, w2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)
        lin = (lambda a, b: ops.linear(a, b, lid)) if hand else (lambda a, b: a @ b)
        h = ops.silu(lin(x, w1), lid) if hand else torch.nn.functional.silu(lin(x, w1))
        w = lin(h, w2)
        if hand:
            t = ops.weighted_channels(sh, w, u, l_max, lid)
        else:
            wr = w.reshape(E, u, l_max + 1)
            t = sh.unsqueeze(1) * torch.cat([wr[:, :, l:l + 1].expand(-1, -1, 2 * l + 1) for l in range(l_max + 1)], dim=-1)
        e = (t * c).sum() + (t ** 2).sum() * 0.1
        gx, gsh = torch.autograd.grad(e, [x, sh], create_graph=True)
        loss = (gx ** 2).sum() + (gsh * sh).sum()
        return [e.detach(), gx.detach(), gsh.detach()] + [v.detach() for v in torch.autograd.grad(loss, [w1, w2, x, sh])]

    for a, b in zip(run(True), run(False)):
        assert (a - b).abs().max().item() <= 1e-10 * max(1.0, float(b.abs().max()))
    # shared weights (`weight_individual_irreps=False`) and l_max = 3
    for l_max_, shared in ((3, False), (2, True), (1, True)):
        D_ = (l_max_ + 1) ** 2
        sh = torch.randn(E, D_, generator=g, dtype=dtype).to(dev).requires_grad_(True)
        w = torch.randn(E, u if shared else u * (l_max_ + 1), generator=g, dtype=dtype).to(dev).requires_grad_(True)
        t = ops.weighted_channels(sh, w, u, l_max_, lid)
        if shared:
            ref = w.unsqueeze(-1) * sh.unsqueeze(1)
        else:
            wr = w.reshape(E, u, l_max_ + 1)
            ref = sh.unsqueeze(1) * torch.cat([wr[:, :, l:l + 1].expand(-1, -1, 2 * l + 1) for l in range(l_max_ + 1)], dim=-1)
        assert (t - ref).abs().max().item() <= 1e-13
        cc = torch.randn(E, u, D_, generator=g, dtype=dtype).to(dev)
        for a, b in zip(torch.autograd.grad((t * cc).sum(), [sh, w]), torch.autograd.grad((ref * cc).sum(), [sh, w])):
            assert (a - b).abs().max().item() <= 1e-12
        # the fused forms against the single ones: one pass over t for both contractions, one store stream for a sum of two products
        meta = (u, l_max_, shared, lid)
        s1, w1 = torch.ops.allegro_amd.weighted_channels_pair(cc, sh.detach(), w.detach(), *meta)
        assert torch.equal(s1, torch.ops.allegro_amd.weighted_channels(2, cc, w.detach(), *meta))
        assert torch.equal(w1, torch.ops.allegro_amd.weighted_channels(1, cc, sh.detach(), *meta))
        sh2, w2 = torch.randn_like(sh), torch.randn_like(w)
        both = torch.ops.allegro_amd.weighted_channels_sum(sh.detach(), w.detach(), sh2, w2, *meta)
        want = torch.ops.allegro_amd.weighted_channels(0, sh.detach(), w.detach(), *meta) + torch.ops.allegro_amd.weighted_channels(0, sh2, w2, *meta)
        assert (both - want).abs().max().item() <= 1e-13
    # rows longer than the register-slot form holds (u D > 2048): the general forward kernel, both with one and two terms
    u_, l_ = 160, 3
    sh = torch.randn(5, 16, generator=g, dtype=dtype).to(dev)
    w = torch.randn(5, u_ * 4, generator=g, dtype=dtype).to(dev)
    sh2, w2 = torch.randn_like(sh), torch.randn_like(w)
    ref = lambda a, b: a.unsqueeze(1) * torch.cat([b.reshape(5, u_, 4)[:, :, l:l + 1].expand(-1, -1, 2 * l + 1) for l in range(4)], dim=-1)  # noqa: E731
    assert (torch.ops.allegro_amd.weighted_channels(0, sh, w, u_, l_, False, lid) - ref(sh, w)).abs().max().item() <= 1e-13
    assert (torch.ops.allegro_amd.weighted_channels_sum(sh, w, sh2, w2, u_, l_, False, lid) - ref(sh, w) - ref(sh2, w2)).abs().max().item() <= 1e-13
    # ... and more than one 64-channel block per edge in the fused pair (160 = 64 + 64 + 32)
    tt = torch.randn(5, u_, 16, generator=g, dtype=dtype).to(dev)
    s1, w1 = torch.ops.allegro_amd.weighted_channels_pair(tt, sh, w, u_, l_, False, lid)
    assert (s1 - torch.ops.allegro_amd.weighted_channels(2, tt, w, u_, l_, False, lid)).abs().max().item() <= 1e-12
    assert torch.equal(w1, torch.ops.allegro_amd.weighted_channels(1, tt, sh, u_, l_, False, lid))


def _silu_case(lib, dev, kind="silu"):
    """Every member of the activation family (SiLU; mish / gelu: `aa_act_derivative`) against autograd through torch's own function, to
    third order, incl. the pair form."""
    lid = _lib_id(lib)
    g = torch.Generator().manual_seed(11)
    ref = {"silu": torch.nn.functional.silu, "mish": torch.nn.functional.mish, "gelu": torch.nn.functional.gelu}[kind]
    hand_fn = (lambda t: ops.silu(t, lid)) if kind == "silu" else (lambda t: ops.activation(t, kind, lid))
    for dtype, tol in ((torch.float64, 1e-12 if kind == "silu" else 1e-11), (torch.float32, 2e-5)):
        for n in (1, 7, 1030):
            x = (3.0 * torch.randn(n, generator=g, dtype=dtype)).to(dev).requires_grad_(True)
            a = torch.randn(n, generator=g, dtype=dtype).to(dev).requires_grad_(True)
            b = torch.randn(n, generator=g, dtype=dtype).to(dev)
            outs = []
            for hand in (True, False):
                y = hand_fn(x) if hand else ref(x)
                (d1,) = torch.autograd.grad((y * a).sum(), x, create_graph=True)  # a f1(x)
                d2x, d2a = torch.autograd.grad((d1 * b).sum(), [x, a], create_graph=True)  # a b f2(x), b f1(x)
                (d3,) = torch.autograd.grad(d2x.sum(), x)  # a b f3(x)
                outs.append([y.detach(), d1.detach(), d2x.detach(), d2a.detach(), d3])
            for p, q in zip(*outs):
                assert (p - q).abs().max().item() <= tol * max(1.0, float(q.abs().max())), (dtype, n)
            # the pair form (what the backward pass of the loss runs: no further derivative recorded)
            y = hand_fn(x)
            (d1,) = torch.autograd.grad((y * a).sum(), x, create_graph=True)
            px, pa = torch.autograd.grad((d1 * b).sum(), [x, a])
            assert (px - outs[1][2]).abs().max().item() <= tol * max(1.0, float(outs[1][2].abs().max()))
            assert (pa - outs[1][3]).abs().max().item() <= tol * max(1.0, float(outs[1][3].abs().max()))


def _linear_forward_case(lib, dev):
    """`aa_linear_forward` (device weights packed per call, split-bf16 matrix-core kernel) against an fp64 product: plain, transposed
    weight view, row-strided operand; and the library-GEMM fallback of `ops.linear` for shapes outside its rules."""
    lid = _lib_id(lib)
    g = torch.Generator().manual_seed(5)
    for E, K, N, pad, transposed in ((300, 64, 192, 0, False), (513, 192, 64, 64, False), (129, 256, 64, 0, True), (77, 32, 32, 0, True)):
        xb = torch.randn(E, K + pad, generator=g).to(dev)
        x = xb[:, :K]
        W = (torch.randn(N, K, generator=g).to(dev).t() if transposed else torch.randn(K, N, generator=g).to(dev))
        got = torch.ops.allegro_amd.linear_forward(x, W, lid)
        want = x.double() @ W.double()
        assert got.shape == (E, N) and (got.double() - want).abs().max().item() <= 3e-6 * K ** 0.5 * float(want.abs().max()), (E, K, N)
    # through the autograd pair, second order, against plain matmuls (fp32 tolerances)
    x0 = torch.randn(200, 64, generator=g).to(dev)
    W0 = torch.randn(64, 96, generator=g).to(dev)
    res = []
    for hand in (True, False):
        x, W = x0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
        y = ops.linear(x, W, lid) if hand else x @ W
        (gx,) = torch.autograd.grad((y ** 2).sum(), x, create_graph=True)
        res.append([y.detach(), gx.detach()] + [v for v in torch.autograd.grad((gx ** 2).sum(), [x, W])])
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, float(b.abs().max()))


def test_linear_forward_emulated():
    from tests.hip_utils import emu_lib

    _linear_forward_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_linear_forward_on_gpu():
    _linear_forward_case(None, torch.device("cuda:0"))


def _fork_cat_case(lib, dev):
    """`ops.fork_scalars` (tensor feature -> itself + its scalar components, gradients merged in one pass) and `ops.cat_features`
    (concatenation with a split as gradient) against slicing / torch.cat, to second order; weights as a row-strided column block."""
    lid = _lib_id(lib)
    dtype = torch.float64
    g = torch.Generator().manual_seed(21)
    E, u, l_max = 29, 6, 2
    D = (l_max + 1) ** 2
    sh0 = torch.randn(E, D, generator=g, dtype=dtype).to(dev)
    wide0 = torch.randn(E, 5 + u * (l_max + 1), generator=g, dtype=dtype).to(dev)  # [scalars | env weights]: w is a strided view
    c = torch.randn(E, u, D, generator=g, dtype=dtype).to(dev)
    Wm = torch.randn(5 + u, 4, generator=g, dtype=dtype).to(dev)
    res = []
    for hand in (True, False):
        sh, wide = sh0.clone().requires_grad_(True), wide0.clone().requires_grad_(True)
        first, w = wide[:, :5], wide[:, 5:]
        t = ops.weighted_channels(sh, w, u, l_max, lid)
        if hand:
            t, s0 = ops.fork_scalars(t, lid)
            feats = ops.cat_features([first, s0], lid)
        else:
            s0 = t[:, :, 0]
            feats = torch.cat([first, s0], dim=-1)
        e = (t * c).sum() + ((feats @ Wm) ** 3).sum() + (t ** 2).sum() * 0.05
        gsh, gw = torch.autograd.grad(e, [sh, wide], create_graph=True)
        loss = (gsh ** 2).sum() + (gw ** 2).sum()
        res.append([e.detach(), gsh.detach(), gw.detach()] + [v for v in torch.autograd.grad(loss, [sh, wide])])
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 1e-10 * max(1.0, float(b.abs().max()))
    # the concatenation kernel itself: row-strided views, widths that allow 16-byte accesses and widths that do not; split_columns
    for widths, dt in (((64, 192), torch.float32), ((5, 6, 1), torch.float64), ((4, 8, 12, 4), torch.float32), ((2, 6), torch.float64)):
        wide = torch.randn(31, sum(widths) + 8, generator=g, dtype=dt).to(dev)
        parts, o = [], 4 if widths[0] % 4 == 0 else 3
        for wd in widths:
            parts.append(wide[:, o:o + wd])
            o += wd
        assert torch.equal(torch.ops.allegro_amd.concat_columns(parts, lid), torch.cat(parts, dim=1))
    x = torch.randn(17, 20, generator=g, dtype=dtype).to(dev).requires_grad_(True)
    a, b, c3 = ops.split_columns(x, [8, 8, 4], lid)
    (gx,) = torch.autograd.grad((a * 2).sum() + (c3 ** 2).sum(), x)  # (b unused: its gradient block is zeros)
    want = torch.cat([torch.full((17, 8), 2.0, dtype=dtype, device=dev), torch.zeros(17, 8, dtype=dtype, device=dev), 2 * x.detach()[:, 16:]], dim=1)
    assert torch.equal(gx, want)
    # the padded form alone (no other gradient of the feature)
    t = torch.randn(E, u, D, generator=g, dtype=dtype).to(dev).requires_grad_(True)
    _, s0 = ops.fork_scalars(t, lid)
    (gt,) = torch.autograd.grad((s0 ** 2).sum(), t)
    want = torch.zeros_like(t)
    want[:, :, 0] = 2 * t.detach()[:, :, 0]
    assert torch.equal(gt, want)
    # features of a pruned layer: any row length (here 13 = 1 + 3 + 9 ... components)
    t = torch.randn(7, u, 13, generator=g, dtype=dtype).to(dev).requires_grad_(True)
    t2, s0 = ops.fork_scalars(t, lid)
    (gt,) = torch.autograd.grad((s0 ** 2).sum() + t2.sum(), t)
    want = torch.ones_like(t)
    want[:, :, 0] += 2 * t.detach()[:, :, 0]
    assert (gt - want).abs().max().item() <= 1e-14


def test_fork_and_cat_emulated():
    from tests.hip_utils import emu_lib

    _fork_cat_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_fork_and_cat_on_gpu():
    _fork_cat_case(None, torch.device("cuda:0"))


@pytest.mark.parametrize("kind", ["mish", "gelu"])
def test_mish_gelu_family_emulated(kind):
    from tests.hip_utils import emu_lib

    _silu_case(emu_lib(), torch.device("cpu"), kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mish", "gelu"])
def test_mish_gelu_family_on_gpu(kind):
    _silu_case(None, torch.device("cuda:0"), kind)


def test_silu_family_emulated():
    from tests.hip_utils import emu_lib

    _silu_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_silu_family_on_gpu():
    _silu_case(None, torch.device("cuda:0"))


def test_linear_wgrad_emulated():
    from tests.hip_utils import emu_lib

    _wgrad_case(emu_lib(), torch.device("cpu"))


def test_training_ops_second_order_emulated():
    from tests.hip_utils import emu_lib

    _second_order_case(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
def test_linear_wgrad_on_gpu():
    _wgrad_case(None, torch.device("cuda:0"))


@pytest.mark.gpu
def test_training_ops_second_order_on_gpu():
    _second_order_case(None, torch.device("cuda:0"))


@pytest.mark.gpu
def test_linear_wgrad_at_c3_size_on_gpu():
    """[298 144 x 64]^T @ [298 144 x 192] in fp32 against an fp64 product: the slab partition at the size the training bench runs."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(298144, 64, device=dev, generator=g)
    gg = torch.randn(298144, 192, device=dev, generator=g)
    got = torch.ops.allegro_amd.linear_wgrad(x, gg, 0)
    want = x.double().t() @ gg.double()
    # (entries are ~ +-2000; fp32 accumulation: 291-term chains inside a slab, then 1024 slabs in order: <= ~1024 x 6e-8 x 600)
    assert (got.double() - want).abs().max().item() <= 0.2
