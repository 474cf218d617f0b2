"""DORMANT leaf-convention check (SURVEY.md section 8c "what can pin the shimmed leaves", Appendix A).

The oracle's parity is pinned at the level of the reference repository: the reference's own files run verbatim, but on a
SHIM of its third-party leaves (e3nn, nequip), restated from the published algorithms and pinned by identities only
(tests/test_conventions.py).  This image has neither package and no network, so every test here SKIPS -- the day they are
importable (`pip install e3nn nequip`, README "Pinning the leaves"), the same command

    python -m pytest tests/test_real_leaves.py -q

compares the shim with the real packages leaf by leaf (real-spherical-harmonic basis order / sign / normalisation, Wigner 3j
in the real basis, Irreps bookkeeping, ScalarMLPFunction parameter names / shapes / scaling constants, Bessel basis and
polynomial cutoff) and runs golden fixtures through the real `allegro` import WITHOUT the shim: their committed outputs were
produced on the shim, so agreement there pins the whole stack."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402

REAL = ref_loader.real_leaves()
need_e3nn = pytest.mark.skipif(REAL["e3nn"] is None, reason="real e3nn not installed (no network in this image): shim leaves stay pinned by identities")
need_nequip = pytest.mark.skipif(REAL["nequip"] is None or REAL["e3nn"] is None, reason="real nequip / e3nn not installed")


def test_ref_loader_prefers_real_packages_over_the_shim(monkeypatch):
    """Runs everywhere: where no real leaf exists the shim goes in front; where one exists it goes LAST (real packages win)."""
    keep = list(sys.path)
    try:
        monkeypatch.delenv("AA_ORACLE_FORCE_SHIM", raising=False)
        ref_loader.install_shim()
        have_real = any(REAL[k] for k in ("e3nn", "nequip"))
        assert sys.path.index(ref_loader.SHIM_ROOT) == (len(sys.path) - 1 if have_real else 0)
        monkeypatch.setenv("AA_ORACLE_FORCE_SHIM", "1")
        ref_loader.install_shim()
        assert sys.path[0] == ref_loader.SHIM_ROOT
        # the shim itself is never mistaken for an installed package
        assert all(v is None or not v.startswith(ref_loader.SHIM_ROOT) for v in ref_loader.real_leaves().values())
    finally:
        sys.path[:] = keep


def test_shim_packages_load_under_an_alias():
    """The comparison below needs shim and real leaves side by side: the shim's e3nn loads as `aa_shim_e3nn`."""
    sh = ref_loader.load_shim_package("e3nn")
    assert sh.__name__ == "aa_shim_e3nn" and sh.o3.Irreps("2x0e+1x1o").dim == 5
    assert tuple(sh.o3.wigner_3j(1, 1, 1).shape) == (3, 3, 3)


@need_e3nn
def test_wigner_3j_and_irreps_match_real_e3nn():
    real = importlib.import_module("e3nn.o3")
    shim = ref_loader.load_shim_package("e3nn").o3
    for l1 in range(4):
        for l2 in range(4):
            for l3 in range(abs(l1 - l2), min(l1 + l2, 3) + 1):
                a = real.wigner_3j(l1, l2, l3, dtype=torch.float64)
                b = shim.wigner_3j(l1, l2, l3, dtype=torch.float64)
                assert torch.allclose(a, b, atol=1e-12), f"wigner_3j({l1},{l2},{l3}): max diff {(a - b).abs().max():.3e} (sign / basis convention)"
    for s in ("64x0e+64x1o+64x2e", "0e + 0o + 1e + 1o", "2o + 1e + 0e", "8x0e+8x1o+8x2e+8x3o"):
        a, b = real.Irreps(s), shim.Irreps(s)
        assert str(a) == str(b) and a.dim == b.dim and a.num_irreps == b.num_irreps and a.lmax == b.lmax
        assert [(m, ir.l, ir.p) for m, ir in a] == [(m, ir.l, ir.p) for m, ir in b]
        assert [(sl.start, sl.stop) for sl in a.slices()] == [(sl.start, sl.stop) for sl in b.slices()]
    assert str(real.Irreps.spherical_harmonics(3, p=-1)) == str(shim.Irreps.spherical_harmonics(3, p=-1))


@need_e3nn
def test_spherical_harmonics_match_real_e3nn():
    """The tensor embedding's call (allegro/nn/tensorembed.py:56-58,85-96): SphericalHarmonics(irreps, normalize=True,
    normalization="component") on edge vectors; basis order and signs within each l are the convention Appendix A restates."""
    real = importlib.import_module("e3nn.o3")
    shim = ref_loader.load_shim_package("e3nn").o3
    g = torch.Generator().manual_seed(5)
    vec = torch.randn(257, 3, generator=g, dtype=torch.float64) * 2.0
    for lmax in (1, 2, 3):
        ir = real.Irreps.spherical_harmonics(lmax, p=-1)
        a = real.SphericalHarmonics(ir, True, "component")(vec)
        b = shim.SphericalHarmonics(shim.Irreps(str(ir)), True, "component")(vec)
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-12), f"l_max {lmax}: max diff {(a - b).abs().max():.3e}"
    a = real.spherical_harmonics([0, 1, 2, 3], vec, True, "component")
    b = shim.spherical_harmonics([0, 1, 2, 3], vec, True, "component")
    assert torch.allclose(a, b, atol=1e-12)


@need_nequip
def test_scalar_mlp_function_matches_real_nequip():
    """Parameter names, shapes, and the forward value for the SAME parameters (i.e. the alpha_i scaling: normalize2mom constant of
    the activation over sqrt(fan_in | fan_out)) -- the constructor arguments are the ones the reference passes
    (allegro/nn/_allegro.py:192-213, allegro_models.py:173-183)."""
    real = importlib.import_module("nequip.nn")
    shim = ref_loader.load_shim_package("nequip").nn
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for nonlin in ("silu", "mish", "gelu", None):
            for fwd_init in (True, False):
                for depth in (0, 1, 2):
                    kw = dict(input_dim=24, output_dim=40, hidden_layers_depth=depth, hidden_layers_width=32, nonlinearity=nonlin,
                              bias=False, forward_weight_init=fwd_init)
                    a, b = real.ScalarMLPFunction(**kw).double(), shim.ScalarMLPFunction(**kw).double()
                    sa, sb = a.state_dict(), b.state_dict()
                    assert list(sa) == list(sb), (list(sa), list(sb))
                    assert [tuple(v.shape) for v in sa.values()] == [tuple(v.shape) for v in sb.values()]
                    b.load_state_dict(sa)
                    x = torch.randn(50, 24, generator=g, dtype=torch.float64)
                    assert torch.allclose(a(x), b(x), rtol=1e-10, atol=1e-12), (nonlin, fwd_init, depth, float((a(x) - b(x)).abs().max()))


@need_nequip
def test_bessel_basis_and_cutoff_match_real_nequip():
    """BesselEdgeLengthEncoding x PolynomialCutoff on normalised lengths (allegro/nn/scalarembed.py:60-66): the stored roots
    (`bessel_weights`: n*pi or n -- aa_model_config.bessel_convention) and the basis values."""
    real_e = importlib.import_module("nequip.nn.embedding")
    real_d = importlib.import_module("nequip.data").AtomicDataDict
    shim_n = ref_loader.load_shim_package("nequip")
    shim_e, shim_d = importlib.import_module("aa_shim_nequip.nn.embedding"), shim_n.data.AtomicDataDict
    x = torch.linspace(0.05, 0.999, 97, dtype=torch.float64).unsqueeze(-1)
    for p in (6, 5.0):
        a = real_e.BesselEdgeLengthEncoding(cutoff=real_e.PolynomialCutoff(p), num_bessels=8, trainable=False).double()
        b = shim_e.BesselEdgeLengthEncoding(cutoff=shim_e.PolynomialCutoff(p), num_bessels=8, trainable=False).double()
        wa, wb = a.bessel_weights.reshape(-1), b.bessel_weights.reshape(-1)
        print("real nequip bessel_weights:", wa.tolist())
        oa = a({real_d.NORM_LENGTH_KEY: x.clone()})
        ob = b({shim_d.NORM_LENGTH_KEY: x.clone()})
        ka, kb = real_d.EDGE_EMBEDDING_KEY, shim_d.EDGE_EMBEDDING_KEY
        assert torch.allclose(oa[real_d.EDGE_CUTOFF_KEY], ob[shim_d.EDGE_CUTOFF_KEY], atol=1e-12)
        # the two published forms differ by the stored roots (n*pi vs n) and agree in VALUE up to the convention the product
        # handles (include/allegro_amd.h: bessel_convention); the shim follows the n*pi form
        assert torch.allclose(oa[ka], ob[kb], atol=1e-10) or torch.allclose(wa * torch.pi, wb, atol=1e-10), \
            "Bessel basis differs from both conventions the product knows"


@need_nequip
@pytest.mark.parametrize("name", ["c2", "t_coupled", "t_spline", "c5_small"])
def test_golden_fixture_through_the_real_packages(name):
    """The reference model built on the REAL nequip / e3nn (no shim in front) loads the committed state_dict (parameter names and
    shapes) and reproduces the committed outputs, which were generated on the shim."""
    if not ref_loader.reference_available():
        pytest.skip("the reference `allegro` package is neither mounted at /root/reference nor installed")
    from tests.golden_utils import load_model_fixture

    ref_loader.import_reference()
    from allegro.model import AllegroModel
    from nequip.data import AtomicDataDict as ADD

    fx = load_model_fixture(name, torch.float64)
    model = AllegroModel(model_dtype="float64", **fx["cfg"]).eval()
    sd = model.state_dict()
    prefix = os.path.commonprefix(list(sd))
    prefix = prefix[: prefix.rfind(".") + 1]
    assert {k[len(prefix):] for k in sd} == set(fx["sd"]), "state_dict keys of the real stack differ from the committed fixture"
    model.load_state_dict({prefix + k: v for k, v in fx["sd"].items()})
    data = {ADD.POSITIONS_KEY: fx["pos"], ADD.EDGE_INDEX_KEY: fx["edge_index"], ADD.ATOM_TYPE_KEY: fx["types"]}
    if fx["shift_vec"] is not None:
        z = np.load(os.path.join(ROOT, "tests", "golden", f"model_{name}.npz"))
        # fixtures store the cartesian shift; every fixture cell is cubic: recover cell and integer shifts
        sv = z["shift_vec"]
        box = float(np.abs(sv[np.abs(sv) > 1e-9]).min()) if (np.abs(sv) > 1e-9).any() else 1.0
        data[ADD.CELL_KEY] = torch.eye(3, dtype=torch.float64) * box
        data[ADD.EDGE_CELL_SHIFT_KEY] = torch.tensor(np.round(sv / box), dtype=torch.float64)
    out = model(data)
    for k in ("atomic_energy", "forces"):
        want = fx["out"][k]
        got = out[k].detach().reshape(want.shape)
        assert (got - want).abs().max().item() <= 1e-9 * max(1.0, float(want.abs().max())), k
