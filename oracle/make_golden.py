"""ORACLE fixture generator (test infrastructure; runs ONLY in the build container).

Imports the reference's own modules verbatim from /root/reference (behind oracle/shim) and dumps
golden input/output vectors into tests/golden/.  The reference cannot travel to the GPU box, so the
vectors are committed together with this script (task rule 3).  Run:  python -m oracle.make_golden

Model fixtures (`model_<name>.npz`): cfg (json), pos/edge_index/types/shift_vec, the reference
state_dict (fp64, "func." prefix stripped), and reference outputs in fp64 and fp32
(atomic_energy, total_energy, forces).
Op fixtures (`contract_cases.npz`): the shapes of the reference's own kernel test
(tests/nn/test_contract_kernels.py:37-40,93-97): 17 edges, 5 atoms, random scatter idxs, mul 3/8,
both weight modes; forward, both input gradients and the path-weight gradient from the reference's eager `Contracter`.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_loader import import_reference  # noqa: E402
from allegro_amd import graph as G  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

BESSEL = {"_target_": "allegro.nn.TwoBodyBesselScalarEmbed", "num_bessels": 8, "polynomial_cutoff_p": 6}
SPLINE = {"_target_": "allegro.nn.TwoBodySplineScalarEmbed", "num_splines": 8, "spline_span": 6}  # test_allegro.py:60-64


def si_cfg(l_max, L, u, S=64, H=64):
    """Hyper-parameters of configs/tutorial.yaml:84-139 (single species Si, no pair potential)."""
    return dict(seed=456, type_names=["Si"], r_max=5.0, l_max=l_max, parity=True, num_layers=L,
                num_scalar_features=S, num_tensor_features=u, radial_chemical_embed=dict(BESSEL),
                radial_chemical_embed_dim=S, scalar_embed_mlp_hidden_layers_depth=1,
                scalar_embed_mlp_hidden_layers_width=H, allegro_mlp_hidden_layers_depth=1,
                allegro_mlp_hidden_layers_width=H, readout_mlp_hidden_layers_depth=1,
                readout_mlp_hidden_layers_width=H, avg_num_neighbors=28.0, tp_path_channel_coupling=True)


def test_cfg(coupling=True, per_edge=False):
    """COMMON_CONFIG of tests/model/test_allegro.py:27-44 (+ minimal_config1 :49-52)."""
    c = dict(seed=123, type_names=["H", "C", "O"], r_max=4.0, avg_num_neighbors=20.0, radial_chemical_embed_dim=16,
             scalar_embed_mlp_hidden_layers_depth=1, scalar_embed_mlp_hidden_layers_width=32, num_layers=2, l_max=2,
             num_scalar_features=32, num_tensor_features=4, allegro_mlp_hidden_layers_depth=2,
             allegro_mlp_hidden_layers_width=32, readout_mlp_hidden_layers_depth=1, readout_mlp_hidden_layers_width=8,
             radial_chemical_embed=dict(BESSEL), parity=True, tp_path_channel_coupling=coupling)
    if per_edge:
        c["per_edge_type_cutoff"] = {"H": 2.0, "C": {"H": 4.0, "C": 3.5, "O": 3.7}, "O": 3.9}
    return c


def water_cfg():
    """Small stand-in for BASELINE config 5: l_max=3, 3 layers, 2 species (fp64)."""
    return dict(seed=7, type_names=["O", "H"], r_max=4.0, l_max=3, parity=True, num_layers=3, num_scalar_features=16,
                num_tensor_features=8, radial_chemical_embed=dict(BESSEL), radial_chemical_embed_dim=16,
                scalar_embed_mlp_hidden_layers_depth=1, scalar_embed_mlp_hidden_layers_width=16,
                allegro_mlp_hidden_layers_depth=1, allegro_mlp_hidden_layers_width=16,
                readout_mlp_hidden_layers_depth=1, readout_mlp_hidden_layers_width=16, avg_num_neighbors=20.0,
                tp_path_channel_coupling=True, per_type_energy_scales=[1.5, 0.7], per_type_energy_shifts=[-3.0, -0.5])


def molecule_graph(n=24, box=9.0, r_cut=4.0, seed=3):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, box, size=(n, 3))
    cell = np.eye(3) * box
    ei, shift = G.neighbor_list_pbc(pos, cell, r_cut)
    return G.Graph(pos=pos, types=rng.integers(0, 3, size=n).astype(np.int64), edge_index=ei, cell=cell,
                   cell_shift=shift, n_local=n)


def run_reference(cfg, g: G.Graph, dtype):
    from allegro.model import AllegroModel
    from nequip.data import AtomicDataDict as ADD

    # weights are always drawn in fp64; the fp32 model is built with model_dtype=float32 (so that
    # tensorembed.py:83,92 picks the fp32 output dtype) and loads the cast fp64 state_dict
    model = AllegroModel(model_dtype="float64", **cfg).eval()
    if dtype == torch.float32:
        m32 = AllegroModel(model_dtype="float32", **cfg).eval()
        m32.load_state_dict({k: v.to(torch.float32) for k, v in model.state_dict().items()})
        model = m32
    data = {ADD.POSITIONS_KEY: torch.tensor(g.pos, dtype=dtype), ADD.EDGE_INDEX_KEY: torch.tensor(g.edge_index),
            ADD.ATOM_TYPE_KEY: torch.tensor(g.types)}
    if g.cell_shift is not None:
        data[ADD.CELL_KEY] = torch.tensor(g.cell, dtype=dtype)
        data[ADD.EDGE_CELL_SHIFT_KEY] = torch.tensor(g.cell_shift, dtype=dtype)
    out = model(data)
    return model, {k: out[k].detach().numpy() for k in ("atomic_energy", "total_energy", "forces")}


def dump_model(name, cfg, g):
    model, out64 = run_reference(cfg, g, torch.float64)
    _, out32 = run_reference(cfg, g, torch.float32)
    arrays = {"cfg_json": np.array(json.dumps(cfg)), "pos": g.pos, "edge_index": g.edge_index, "types": g.types}
    if g.cell_shift is not None:
        arrays["shift_vec"] = g.shift_vec()
    for k, v in model.state_dict().items():
        assert k.startswith("func.")
        arrays["sd/" + k[len("func."):]] = v.detach().to(torch.float64).numpy()
    for k, v in out64.items():
        arrays["out64/" + k] = v
    for k, v in out32.items():
        arrays["out32/" + k] = v
    path = os.path.join(GOLD, f"model_{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: N={g.num_atoms} E={g.num_edges} E_tot={float(out64['total_energy'].sum()):.10f} "
          f"|F|max={np.abs(out64['forces']).max():.6f}  f32-f64 dF={np.abs(out32['forces'] - out64['forces']).max():.2e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KB")


def dump_ghost(name, base, cfg, g):
    """`pair_allegro` tensor contract from the reference's OWN data transform: `allegro_data_settings`
    (allegro/_compile.py:17-65) turns the periodic frame into the ghost-atom layout LAMMPS hands over -- ghosts
    appended, no cell, edges in ITS order (inside-cell edges first, then the outside-cell ones, i.e. NOT sorted by
    center, :47-58) -- and the reference model runs on the result.  Stored: the transformed inputs exactly as emitted
    and the reference's outputs on them (ghost rows carry the force contributions LAMMPS reverse-communicates).
    The weights are those of fixture `base` (same cfg and seed; asserted)."""
    from allegro._compile import PAIR_ALLEGRO_INPUTS, allegro_data_settings
    from allegro.model import AllegroModel
    from nequip.data import AtomicDataDict as ADD

    arrays = {"cfg_json": np.array(json.dumps(cfg)), "weights_of": np.array(base), "n_local": np.array(g.num_atoms)}
    zb = np.load(os.path.join(GOLD, f"model_{base}.npz"))
    for dtype, tag in ((torch.float64, "64"), (torch.float32, "32")):
        model = AllegroModel(model_dtype="float64", **cfg).eval()
        for k, v in model.state_dict().items():
            assert np.array_equal(zb["sd/" + k[len("func."):]], v.detach().to(torch.float64).numpy()), k
        if dtype == torch.float32:
            m32 = AllegroModel(model_dtype="float32", **cfg).eval()
            m32.load_state_dict({k: v.to(torch.float32) for k, v in model.state_dict().items()})
            model = m32
        data = {ADD.POSITIONS_KEY: torch.tensor(g.pos, dtype=dtype), ADD.EDGE_INDEX_KEY: torch.tensor(g.edge_index),
                ADD.ATOM_TYPE_KEY: torch.tensor(g.types), ADD.CELL_KEY: torch.tensor(g.cell, dtype=dtype),
                ADD.EDGE_CELL_SHIFT_KEY: torch.tensor(g.cell_shift, dtype=dtype)}
        data = allegro_data_settings(data)  # the reference's transform, verbatim
        assert ADD.CELL_KEY not in data and ADD.EDGE_CELL_SHIFT_KEY not in data
        if tag == "64":
            assert PAIR_ALLEGRO_INPUTS == [ADD.POSITIONS_KEY, ADD.EDGE_INDEX_KEY, ADD.ATOM_TYPE_KEY]
            arrays["pos"] = data[ADD.POSITIONS_KEY].numpy()
            arrays["edge_index"] = data[ADD.EDGE_INDEX_KEY].numpy()
            arrays["types"] = data[ADD.ATOM_TYPE_KEY].numpy()
            c = arrays["edge_index"][0]
            assert (np.diff(c) < 0).any(), "the reference emits inside-cell edges first: not center-sorted"
        out = model({k: v for k, v in data.items()})
        for k in ("atomic_energy", "total_energy", "forces"):
            arrays[f"out{tag}/{k}"] = out[k].detach().numpy()
    n = g.num_atoms
    path = os.path.join(GOLD, f"model_{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {n} local + {arrays['pos'].shape[0] - n} ghost atoms, E={arrays['edge_index'].shape[1]} "
          f"E_local={float(arrays['out64/atomic_energy'][:n].sum()):.10f} |F|max={np.abs(arrays['out64/forces']).max():.6f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KB")


def dump_contract_cases():
    from allegro.nn._strided import Contracter
    from e3nn import o3

    torch.set_default_dtype(torch.float64)
    arrays, names = {}, []
    gen = torch.Generator().manual_seed(11)
    idx = 0
    for in1 in ["0e + 0o + 1e + 1o", "2o + 1e + 0e"]:
        for out in ["0e + 0o + 1e + 1o", "1o + 2e"]:
            for coupling in [True, False]:
                for mul in [3, 8]:
                    i1, i2, io = o3.Irreps(in1), o3.Irreps("0e + 0o + 1e + 1o"), o3.Irreps(out)
                    torch.manual_seed(100 + idx)
                    c = Contracter(irreps_in1=i1, irreps_in2=i2, irreps_out=io, mul=mul,
                                   path_channel_coupling=coupling, scatter_factor=0.37)
                    E, N = 17, 5
                    x1 = torch.randn(E, mul, i1.dim, generator=gen, requires_grad=True)
                    x2 = torch.randn(E, mul, i2.dim, generator=gen, requires_grad=True)
                    idxs = torch.randint(0, N, (E,), generator=gen)
                    y = c(x1, x2, idxs, N)
                    gy = torch.randn(y.shape, generator=gen)
                    # input gradients and the path-weight gradient of the reference's eager Contracter (autograd through
                    # its `weights` Parameter, _contract.py:172-177,219)
                    g1, g2, gw = torch.autograd.grad(y, [x1, x2, c.weights], gy)
                    tag = f"case{idx}"
                    names.append(tag)
                    meta = dict(irreps_in1=in1, irreps_in2="0e + 0o + 1e + 1o", irreps_out=out, mul=mul,
                                coupling=coupling, num_atoms=N, scatter_factor=0.37,
                                ij_diagonal=bool(c.w3j_is_ij_diagonal), num_paths=c.num_paths)
                    arrays[f"{tag}/meta"] = np.array(json.dumps(meta))
                    for k, v in dict(x1=x1, x2=x2, idxs=idxs, weights=c.weights, w3j=c.w3j, out=y, gout=gy, gx1=g1,
                                     gx2=g2, gw=gw).items():
                        arrays[f"{tag}/{k}"] = v.detach().numpy()
                    idx += 1
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "contract_cases.npz"), **arrays)
    print(f"contract cases: {idx}")
    torch.set_default_dtype(torch.float32)


def main(only=None):
    """`only`: names of the fixtures to (re)generate (default: all)."""
    import_reference()
    os.makedirs(GOLD, exist_ok=True)
    all_dump = globals()["dump_model"]

    def dump_model(name, cfg, g):
        if not only or name in only:
            all_dump(name, cfg, g)

    si = G.make_si_graph(2)
    dump_model("c1_L1", si_cfg(1, 1, 32), si)  # BASELINE config 0 as glossed ("1 layer")
    dump_model("c1_L2", si_cfg(1, 2, 32), si)  # BASELINE config 0 as configs/tutorial.yaml:103-114 says
    dump_model("c2", si_cfg(2, 2, 64), si)  # BASELINE config 1
    mol = molecule_graph()
    dump_model("t_coupled", test_cfg(True, False), mol)
    dump_model("t_uncoupled", test_cfg(False, False), mol)
    dump_model("t_peredge", test_cfg(True, True), mol)
    # the spline two-body embedding of the reference's model test matrix (test_allegro.py:60-64,75-79)
    dump_model("t_spline", dict(test_cfg(True, False), radial_chemical_embed=dict(SPLINE)), mol)
    dump_model("t_spline_peredge", dict(test_cfg(False, True), radial_chemical_embed=dict(SPLINE)), mol)
    dump_model("c2_spline", dict(si_cfg(2, 2, 64), radial_chemical_embed=dict(SPLINE)), si)
    # the other specialised kernel families, straight from the reference (C2 geometry):
    dump_model("c2_l3", si_cfg(3, 2, 64), si)                      # moments + chains at l_max = 3
    dump_model("c2_l1", si_cfg(1, 2, 64), si)                      # ... at l_max = 1
    dump_model("c2_L3", si_cfg(2, 3, 64), si)                      # 3 layers: per-atom operator path + chains
    dump_model("c2_u128", si_cfg(2, 2, 128, S=128, H=128), si)     # 128 features: operator path, single-layer GEMMs
    dump_model("c2_uncoupled", dict(si_cfg(2, 2, 64), tp_path_channel_coupling=False), si)  # p-mode weights
    # constructor options of the reference beyond the defaults (allegro_models.py:49-60,126-142): the other MLP
    # nonlinearities (incl. None = linear MLPs) and one env weight per channel shared by all irreps
    dump_model("t_acts", dict(test_cfg(True, False), scalar_embed_mlp_nonlinearity="gelu", allegro_mlp_nonlinearity="mish",
                              readout_mlp_nonlinearity=None), mol)
    dump_model("t_mish", dict(test_cfg(False, True), scalar_embed_mlp_nonlinearity="mish", allegro_mlp_nonlinearity="gelu",
                              readout_mlp_nonlinearity="mish"), mol)
    dump_model("t_shared", dict(test_cfg(True, False), weight_individual_irreps=False), mol)
    dump_model("c2_shared", dict(si_cfg(2, 2, 64), weight_individual_irreps=False), si)  # ... on the moments / chain fast path
    for seed in range(100):  # pick a seed without unphysically close intermolecular contacts
        w = G.make_water_graph(3, 9.9, r_cut=4.0, seed=seed)
        r = w.pos[w.edge_index[1]] - w.pos[w.edge_index[0]] + w.shift_vec()
        d = np.sort(np.linalg.norm(r, axis=1))
        if d[2 * 2 * 27] > 1.45:  # skip the 2*27 intramolecular O-H bonds (directed: x2); H-H intra is 1.51
            break
    dump_model("c5_small", water_cfg(), w)
    if not only or "c2_ghost" in only:
        dump_ghost("c2_ghost", "c2", si_cfg(2, 2, 64), si)  # the reference's own pair_allegro data transform on C2
    if not only or "contract_cases" in only:
        dump_contract_cases()


if __name__ == "__main__":
    main(sys.argv[1:])
