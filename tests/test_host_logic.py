"""CPU: host-side logic -- irreps bookkeeping, Wigner 3j, graph construction, ghost layout, partition."""
import numpy as np
import pytest
import torch

from allegro_amd import graph as G
from allegro_amd import o3
from allegro_amd.nn import allegro_layer_irreps, build_w3j


def test_layer_irreps_match_survey_table():
    # SURVEY.md §8 table (derived from allegro/nn/_allegro.py:101-160)
    t = allegro_layer_irreps(2, True, 2)
    assert [x.dim for x in t] == [9, 9, 1]
    t = allegro_layer_irreps(3, True, 3)
    assert [x.dim for x in t] == [16, 31, 16, 1]
    t = allegro_layer_irreps(1, True, 1)
    assert [x.dim for x in t] == [4, 1]


@pytest.mark.parametrize("lmax,L,paths,nnz,diag", [(1, 2, [4, 2], [10, 4], [False, True]),
                                                  (2, 2, [11, 3], [83, 9], [False, True]),
                                                  (3, 3, [34, 34, 4], [611, 611, 16], [False, False, True])])
def test_paths_and_nnz_match_survey_table(lmax, L, paths, nnz, diag):
    t = allegro_layer_irreps(lmax, True, L)
    env = o3.Irreps.spherical_harmonics(lmax)
    for l in range(L):
        w3j, instr, d, dims = build_w3j(t[l], env, t[l + 1])
        assert len(instr) == paths[l] and int((w3j != 0).sum()) == nnz[l] and d == diag[l]


def test_wigner_identities():
    for l in range(4):
        w = o3.wigner_3j(l, l, 0)
        assert np.allclose(w[:, :, 0], np.eye(2 * l + 1) / np.sqrt(2 * l + 1))
    for (a, b, c), n in {(1, 1, 0): 3, (1, 1, 1): 6, (1, 1, 2): 11, (2, 2, 2): 25, (1, 2, 3): 21}.items():
        w = o3.wigner_3j(a, b, c)
        assert abs(np.linalg.norm(w) - 1) < 1e-12 and int((w != 0).sum()) == n


def test_si_box_and_ghost_layout():
    g = G.make_si_graph(2)
    assert g.num_atoms == 64 and g.num_edges == 64 * 28 and G.is_center_sorted(g.edge_index[0])
    r = g.pos[g.edge_index[1]] - g.pos[g.edge_index[0]] + g.shift_vec()
    d = np.linalg.norm(r, axis=1)
    assert d.max() < 5.0
    gg = G.to_ghost_layout(g)
    r2 = gg.pos[gg.edge_index[1]] - gg.pos[gg.edge_index[0]]
    # reference property (tests/utils/test_compile_utils.py:7-18): same multiset of edge lengths
    assert np.allclose(np.sort(np.linalg.norm(r2, axis=1)), np.sort(d))
    assert gg.n_local == 64 and G.is_center_sorted(gg.edge_index[0])


def test_partition_covers_all_edges():
    from allegro_amd.dist import partition_atoms

    g = G.make_si_graph(3)
    rowptr = G.csr_from_sorted_centers(g.edge_index[0], g.num_atoms)
    for parts in (1, 2, 4, 8):
        cuts = partition_atoms(rowptr, parts)
        assert cuts[0] == 0 and cuts[-1] == g.num_atoms and all(a <= b for a, b in zip(cuts, cuts[1:]))
        sizes = [rowptr[b] - rowptr[a] for a, b in zip(cuts, cuts[1:])]
        assert sum(sizes) == g.num_edges and max(sizes) - min(sizes) <= 2 * 28


def test_bench_roofline_accounting_on_synthetic_stages():
    """bench.py's host-side accounting: the dominant symbol, per-launch algorithmic GB/s and the SURVEY 8(d) step
    roofline (424 kflop/edge forward+force at the C2-C4 model) from a synthetic stage list."""
    import bench

    stages = [("gc_64x64_64x64_64x256", 1.0, 4.0e9, 1.0e11), ("gc_128x64_64x64", 0.5, 2.0e9, 0.5e11),
              ("tp_mom_fwd_first", 0.9, 3.6e9, 0.0), ("edge_prologue", 0.3, 0.9e9, 0.0)]
    roof, table = bench.roofline_from_stages(stages, "float32", workload="none")
    assert roof["kernel"] == "gemm_chain_bf16x3_kernel" and roof["launches_per_step"] == 2 and roof["bound"] == "hbm"
    assert abs(roof["achieved"] - 4000.0) < 1e-6 and abs(roof["frac"] - 0.5) < 1e-9 and roof["traffic"] is None
    assert abs(roof["fp32_equiv_TFLOPs"] - 100.0) < 1e-6 and list(table)[0] == "gemm_chain_bf16x3_kernel"
    cfg = bench.si_model_cfg()
    sr = bench.step_roofline(cfg, 1000, 1e-3, stages, "float32")
    assert abs(sr["flop_per_edge"] - 423898.0) < 1.0                     # ~424 kflop/edge
    assert abs(sr["algorithmic_bytes_per_edge"] - 10.5e9 / 1000) < 1e-3


def test_bench_prices_the_fused_forward_against_the_matrix_roofline():
    """The fused forward (one launch, ~80 flop/B) is matrix-bound: fp32-equivalent flops of its linear layers against the
    fp32 MFMA peak, with the HBM rate kept as a secondary figure."""
    import bench

    stages = [("fused_fwd", 5.0, 6.0e9, 3.0e11), ("gc_64x64_64x64_128x128_64x64", 1.4, 4.2e9, 1.5e11), ("tp_mom_bwd_first", 2.0, 7.7e9, 0.0)]
    roof, _ = bench.roofline_from_stages(stages, "float32", workload="none")
    assert roof["kernel"] == "fused_fwd" and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == bench.PEAK_F32_TFLOPS
    assert abs(roof["achieved"] - 60.0) < 1e-9 and abs(roof["frac"] - 60.0 / 157.3) < 1e-9 and abs(roof["hbm_GBps"] - 1200.0) < 1e-6


def test_train_mode_keeps_user_freezes_and_bessel_roots_are_a_parameter_when_trainable():
    """`train()` must not re-enable what the user froze, and `bessel_trainable=True` must register the
    roots as a Parameter (nequip's BesselEdgeLengthEncoding trains them; the training evaluator differentiates through them)."""
    from allegro_amd.nn import HipAllegroModel
    from tests.fastpath_utils import _cfg

    cfg = _cfg("bessel", True)
    m = HipAllegroModel(**cfg)
    key = "func.radial_chemical_embed.bessel_encode.bessel_weights"
    assert key not in dict(m.named_parameters()) and key in dict(m.named_buffers())
    cfg["radial_chemical_embed"] = dict(cfg["radial_chemical_embed"], bessel_trainable=True)
    m = HipAllegroModel(**cfg, bessel_convention="npi")
    assert key in dict(m.named_parameters())
    assert not m.training and not any(p.requires_grad for p in m.parameters())  # inference pipeline by default
    m.train()
    assert all(p.requires_grad for k, p in m.named_parameters() if k not in m._frozen_keys)  # (per-type scales / shifts: only when declared trainable)
    frozen = [k for k, _ in m.named_parameters() if ".edge_readout." in k] + sorted(m._frozen_keys)
    assert frozen
    for k, p in m.named_parameters():
        if k in frozen:
            p.requires_grad_(False)
    m.train()  # (every epoch of a training loop)
    assert not any(p.requires_grad for k, p in m.named_parameters() if k in frozen)
    m.eval()
    assert not any(p.requires_grad for p in m.parameters())
    m.train()
    on = {k for k, p in m.named_parameters() if p.requires_grad}
    assert on == {k for k, _ in m.named_parameters()} - set(frozen)
