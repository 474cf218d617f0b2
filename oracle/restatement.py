"""ORACLE -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product package `allegro_amd` never does.

What it is: a plain-PyTorch (CPU, fp32/fp64) functional restatement of
mir-group/allegro's forward + force evaluation, following SURVEY.md §3.1 step by step,
each function citing the reference file:line it follows.  It consumes the reference
model's own `state_dict` (incl. the persistent `w3j` buffers, _contract.py:168).

How it is pinned: /root/reference holds NO golden vectors, so this file is
pinned against outputs of the reference ITSELF: tests/golden/*.npz were produced by
oracle/make_golden.py, which imports the reference's own modules verbatim from
/root/reference (behind the leaf shim in oracle/shim for the absent e3nn/nequip/hydra
packages) and dumps energies/forces; tests/test_oracle_golden.py checks this restatement
against every such fixture (fp64 1e-10, fp32 5e-5 -- the reference's own model tolerances,
tests/model/test_allegro.py:72-74).  Residual: the e3nn/nequip LEAVES are restated from
memory in the shim -> "parity unpinned" for absolute values at that boundary.
"""
import math
from typing import Dict, Optional

import torch


# ----------------------------------------------------------------------------- leaves
def silu_second_moment_const() -> float:
    """e3nn normalize2mom constant for SiLU (used inside nequip ScalarMLPFunction; Appendix A)."""
    z = torch.linspace(-12.0, 12.0, 240001, dtype=torch.float64)
    w = torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    return 1.0 / math.sqrt(torch.trapezoid(torch.nn.functional.silu(z) ** 2 * w, z).item())


def spherical_harmonics_lmax3(vec: torch.Tensor, l_max: int) -> torch.Tensor:
    """Component-normalised real SH of the normalised vector, y polar, m=-l..l
    (follows the call at allegro/nn/tensorembed.py:55-57,92; explicit polynomials, l<=3)."""
    assert 0 <= l_max <= 3
    n = vec / torch.linalg.norm(vec, dim=-1, keepdim=True)
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    out = [torch.ones_like(x)]
    if l_max >= 1:
        s3 = math.sqrt(3.0)
        out += [s3 * x, s3 * y, s3 * z]
    if l_max >= 2:
        s15, s5 = math.sqrt(15.0), math.sqrt(5.0)
        out += [s15 * x * z, s15 * x * y, s5 * (y * y - 0.5 * (x * x + z * z)), s15 * y * z,
                0.5 * s15 * (z * z - x * x)]
    if l_max >= 3:
        c70, c105, c42, c7 = math.sqrt(70.0) / 4, math.sqrt(105.0), math.sqrt(42.0) / 4, math.sqrt(7.0) / 2
        out += [c70 * x * (3 * z * z - x * x), c105 * x * y * z, c42 * x * (5 * y * y - 1.0),
                c7 * y * (5 * y * y - 3.0), c42 * z * (5 * y * y - 1.0), 0.5 * c105 * y * (z * z - x * x),
                c70 * z * (z * z - 3 * x * x)]
    return torch.stack(out, dim=-1)


def polynomial_cutoff(x: torch.Tensor, p: float) -> torch.Tensor:
    """nequip PolynomialCutoff (scalarembed.py:61; Appendix A)."""
    out = 1.0 - ((p + 1.0) * (p + 2.0) / 2.0) * x**p + p * (p + 2.0) * x ** (p + 1.0) - (p * (p + 1.0) / 2.0) * x ** (p + 2.0)
    return out * (x < 1.0)


_ACTS = {"silu": torch.nn.functional.silu, "mish": torch.nn.functional.mish, "gelu": torch.nn.functional.gelu}
_ACT_CONSTS: Dict[str, float] = {}


def second_moment_const(nonlinearity: Optional[str]) -> float:
    """normalize2mom constant of the MLP nonlinearity (1 for None)."""
    if nonlinearity is None:
        return 1.0
    if nonlinearity not in _ACT_CONSTS:
        z = torch.linspace(-12.0, 12.0, 240001, dtype=torch.float64)
        w = torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
        _ACT_CONSTS[nonlinearity] = 1.0 / math.sqrt(torch.trapezoid(_ACTS[nonlinearity](z) ** 2 * w, z).item())
    return _ACT_CONSTS[nonlinearity]


def scalar_mlp(x: torch.Tensor, weights, forward_weight_init: bool = True, act_const: Optional[float] = None,
               nonlinearity: Optional[str] = "silu"):
    """nequip ScalarMLPFunction: y = x @ (W_i*alpha_i), the nonlinearity (silu / mish / gelu / None,
    allegro_models.py:49-60) between layers (Appendix A).
    Call sites: _allegro.py:90-94,193-213,251,278; tensorembed.py:76-81,89; allegro_models.py:173,231."""
    act_const = second_moment_const(nonlinearity)
    norm_from_last = 1.0
    n = len(weights)
    for i, w in enumerate(weights):
        alpha = norm_from_last / math.sqrt(float(w.shape[0] if forward_weight_init else w.shape[1]))
        x = x @ (w * alpha)
        if i < n - 1 and nonlinearity is not None:
            x = _ACTS[nonlinearity](x)
            norm_from_last = act_const
    return x


def _mlp_weights(sd: Dict[str, torch.Tensor], prefix: str):
    ws, i = [], 0
    while f"{prefix}.{i}.weight" in sd:
        ws.append(sd[f"{prefix}.{i}.weight"])
        i += 1
    assert ws, f"no MLP weights under {prefix}"
    return ws


# ----------------------------------------------------------------------------- strided ops
def make_weighted_channels(sh: torch.Tensor, w: torch.Tensor, u: int, l_max: int) -> torch.Tensor:
    """out[z,u,i] = sh[z,i] * w[z,u,irrep(i)]  (allegro/nn/_strided/_channels.py:44-57; layout [z,u,r]); with
    weight_individual_irreps=False the weights are [z,u] and shared by all irreps (:60-63)."""
    if w.shape[1] == u:
        return w.unsqueeze(-1) * sh.unsqueeze(-2)
    r_of_i = torch.tensor([l for l in range(l_max + 1) for _ in range(2 * l + 1)], device=sh.device)
    wz = w.reshape(sh.shape[0], u, l_max + 1)
    return sh.unsqueeze(1) * wz[:, :, r_of_i]


def contract(x1, x2, weights, w3j, coupling: bool):
    """Contracter._contract (allegro/nn/_strided/_contract.py:213-251) without the [z,u,i,j,k]
    intermediate: ww3j folded first (:219), then a batched matmul over channels."""
    if coupling:
        p = weights.shape[1] if weights.dim() == 2 else 1
    else:
        p = weights.shape[0] if weights.dim() == 1 else 1
    w3 = w3j if p > 1 else w3j.unsqueeze(0)
    wts = weights if p > 1 else weights.unsqueeze(-1)
    diag = w3.dim() == 3  # [p,i,k]
    u = x1.shape[1]
    if coupling:
        ww = torch.einsum("up,p...->u...", wts, w3)  # [u,i,(j,)k]
    else:
        ww = torch.einsum("p,p...->...", wts, w3).unsqueeze(0).expand((u,) + tuple(w3.shape[1:]))
    if diag:
        return torch.einsum("zui,uik->zuk", x1 * x2, ww)
    outer = (x1.unsqueeze(-1) * x2.unsqueeze(-2)).reshape(x1.shape[0], u, -1)
    return torch.einsum("zua,uak->zuk", outer, ww.reshape(u, -1, ww.shape[-1]))


def contracter_forward(x1, x2, idxs, num_atoms, weights, w3j, coupling, scatter_factor):
    """Contracter.forward (allegro/nn/_strided/_contract.py:185-211): scale, scatter-sum by center,
    gather back with the SAME index, then contract."""
    if scatter_factor is not None:
        x2 = scatter_factor * x2
    x2s = torch.zeros((num_atoms,) + tuple(x2.shape[1:]), dtype=x2.dtype, device=x2.device).index_add_(0, idxs, x2)
    x2 = x2s.index_select(0, idxs)
    return contract(x1, x2, weights, w3j, coupling)


# ----------------------------------------------------------------------------- full path
def allegro_energy(cfg: dict, sd: Dict[str, torch.Tensor], pos, edge_index, atom_types, shift_vec=None,
                   return_intermediates: bool = False, contracters=None):
    """Forward of AllegroEnergyModel (module order: allegro/model/allegro_models.py:222-228,262-268,297).
    `sd` keys are the reference state_dict keys with the leading "func." stripped.
    `contracters` (tests of the operator seam): one callable per layer with the signature of
    `Contracter.forward(x1, x2, idxs, scatter_dim_size)` (_contract.py:185) used instead of `contracter_forward`."""
    S, u, L, l_max = cfg["num_scalar_features"], cfg["num_tensor_features"], cfg["num_layers"], cfg["l_max"]
    coupling = cfg.get("tp_path_channel_coupling", True)
    fwi = cfg.get("forward_normalize", True)
    avg_nn = float(cfg["avg_num_neighbors"])
    act_c = silu_second_moment_const()
    center, nbr = edge_index[0], edge_index[1]
    N = pos.shape[0]
    inter = {}
    # 1 edge_norm (allegro_models.py:153-157) + with_edge_vectors_ (tensorembed.py:86)
    vec = pos.index_select(0, nbr) - pos.index_select(0, center)
    if shift_vec is not None:
        vec = vec + shift_vec
    r = torch.linalg.norm(vec, dim=-1)
    recip = sd["edge_norm.rmax_recip"]
    if recip.numel() > 1:
        et = atom_types[edge_index]
        x = (r * recip[et[0], et[1]]).unsqueeze(-1)
    else:
        x = (r * recip.reshape(-1)[0]).unsqueeze(-1)
    et = atom_types[edge_index]
    if "radial_chemical_embed.spline.class_embed.weight" in sd:
        # 2' TwoBodySplineScalarEmbed (scalarembed.py:157-175) -> PerClassSpline.forward (spline.py:64-89)
        lower, upper = sd["radial_chemical_embed.spline.lower"], sd["radial_chemical_embed.spline.upper"]
        ns = lower.numel()
        const = 2 * math.pi / float(upper[0] - lower[0])
        nx = const * (torch.minimum(torch.maximum(x, lower), upper) - lower)
        sbasis = 0.25 * (1 - torch.cos(nx)).square()
        classes = et[0] * len(cfg["type_names"]) + et[1]
        wsp = sd["radial_chemical_embed.spline.class_embed.weight"][classes].view(classes.shape[0], -1, ns)
        emb = torch.bmm(wsp, sbasis.unsqueeze(-1)).squeeze(-1)
    else:
        # 2 radial_chemical_embed: Bessel x cutoff -> ProductTypeEmbedding (scalarembed.py:60-81; _edgeembed.py:68-84)
        bw = sd["radial_chemical_embed.bessel_encode.bessel_weights"]
        # nequip BesselEdgeLengthEncoding (EXT), two published forms, told apart by the stored roots: n*pi with
        # sin(w x)/x (what the shim that generated the golden vectors uses), or n with sinc(x w) w = sin(pi w x)/(pi x)
        if torch.allclose(bw.reshape(-1).double(), torch.arange(1, bw.numel() + 1, dtype=torch.float64, device=bw.device)):
            bessel = torch.sinc(x * bw) * bw
        else:
            bessel = torch.sin(bw * x) / x
        bessel = bessel * polynomial_cutoff(x, float(cfg.get("polynomial_cutoff_p", 6)))
        type_embed = torch.cat((sd["radial_chemical_embed.type_embed.center_embed.weight"][et[0]],
                                sd["radial_chemical_embed.type_embed.neighbor_embed.weight"][et[1]]), dim=-1)
        basis = scalar_mlp(bessel, _mlp_weights(sd, "radial_chemical_embed.type_embed.basis_linear.mlp"), fwi, act_c)
        emb = type_embed * basis
    inter["emb0"] = emb
    # 3 scalar_embed_mlp (allegro_models.py:173-183)
    emb = scalar_mlp(emb, _mlp_weights(sd, "scalar_embed_mlp.mlp.mlp"), fwi, act_c,
                     cfg.get("scalar_embed_mlp_nonlinearity", "silu"))
    inter["edge_embedding"] = emb
    # 4 tensor_embed (tensorembed.py:85-96)
    w0 = scalar_mlp(emb, _mlp_weights(sd, "tensor_embed.env_embed_linear.mlp"), fwi, act_c)
    sh = spherical_harmonics_lmax3(vec, l_max)
    inter["edge_attrs"] = sh
    tf = make_weighted_channels(sh, w0, u, l_max)
    # 5 allegro (_allegro.py:237-301)
    W = (l_max + 1) * u if cfg.get("weight_individual_irreps", True) else u  # env-weight columns (_channels.py:29-35)
    proj = scalar_mlp(emb, _mlp_weights(sd, "allegro.first_layer_env_embed_projection.mlp"), fwi, act_c)
    acc = [proj[:, :S]]
    env_w = proj[:, S:S + W]
    for layer in range(L):
        env = make_weighted_channels(sh, env_w, u, l_max)  # :263
        w3j = sd[f"allegro.tps.{layer}.w3j"]
        d1 = tf.shape[-1]
        # later layers use only the irreps that survived pruning; with parity=True, L<=2 they equal SH irreps
        if contracters is not None:
            tf = contracters[layer](tf.reshape(-1, u, d1), env, center, N)  # :268
        else:
            tf = contracter_forward(tf.reshape(-1, u, d1), env, center, N, sd[f"allegro.tps.{layer}.weights"], w3j,
                                    coupling, 1.0 / math.sqrt(avg_nn))  # :268, _contract.py:185-211
        inter[f"tf{layer + 1}"] = tf
        scalars = tf[:, :, :1].reshape(tf.shape[0], u)  # :272-275
        lat = scalar_mlp(torch.cat(acc + [scalars], dim=-1), _mlp_weights(sd, f"allegro.latents.{layer}.mlp"), fwi, act_c,
                         cfg.get("allegro_mlp_nonlinearity", "silu"))
        acc.append(lat[:, :S])  # :284-286
        if layer < L - 1:
            env_w = lat[:, S:S + W]  # :289-294
    feats = torch.cat(acc, dim=-1)  # :300
    inter["edge_features"] = feats
    # 6 edge_readout (allegro_models.py:231-241), 7 edge_eng_sum (edgewise.py:40-60; factor allegro_models.py:245)
    e_edge = scalar_mlp(feats, _mlp_weights(sd, "edge_readout.mlp.mlp"), fwi, act_c,
                        cfg.get("readout_mlp_nonlinearity", "silu"))
    e_edge = e_edge * (1.0 / math.sqrt(2 * avg_nn))
    e_atom = torch.zeros((N, 1), dtype=e_edge.dtype, device=e_edge.device).index_add_(0, center, e_edge)
    # 8 per_type_energy_scale_shift (allegro_models.py:251-260)
    if "per_type_energy_scale_shift.scales" in sd:
        e_atom = e_atom * sd["per_type_energy_scale_shift.scales"][atom_types].reshape(-1, 1)
    if "per_type_energy_scale_shift.shifts" in sd:
        e_atom = e_atom + sd["per_type_energy_scale_shift.shifts"][atom_types].reshape(-1, 1)
    if return_intermediates:
        return e_atom, inter
    return e_atom


def allegro_energy_forces(cfg, sd, pos, edge_index, atom_types, shift_vec=None):
    """ForceStressOutput(AllegroEnergyModel) (allegro_models.py:101-103): forces = -dE_total/dpos."""
    pos = pos.detach().clone().requires_grad_(True)
    e_atom = allegro_energy(cfg, sd, pos, edge_index, atom_types, shift_vec)
    e_tot = e_atom.sum()
    (g,) = torch.autograd.grad(e_tot, pos)
    return {"atomic_energy": e_atom.detach(), "total_energy": e_tot.detach().reshape(1, 1), "forces": -g}


def allegro_virial(cfg, sd, pos, edge_index, atom_types, shift_vec=None):
    """dE_total/d(strain) [3,3]: positions and periodic shift vectors are displaced by x -> x + x @ eps^T and the
    energy is differentiated at eps = 0 (the strain-displacement construction of nequip's ForceStressOutput -- EXT,
    restated from memory; stress = this / volume).  Test-only checker for aa_model_virial."""
    eps = torch.zeros(3, 3, dtype=pos.dtype, requires_grad=True)
    p2 = pos.detach() + pos.detach() @ eps.T
    s2 = None if shift_vec is None else shift_vec.detach() + shift_vec.detach() @ eps.T
    e_tot = allegro_energy(cfg, sd, p2, edge_index, atom_types, s2).sum()
    (g,) = torch.autograd.grad(e_tot, eps)
    return g


def allegro_energy_forces_chunked(cfg, sd, pos, edge_index, atom_types, shift_vec=None, max_edges: int = 20000):
    """Exact evaluation in contiguous center-atom blocks (strict locality, tests/model/test_allegro.py:68-70):
    every E_i depends only on edges (i, .), so blocks of centers are independent; forces accumulate.
    Requires edges sorted by center."""
    N = pos.shape[0]
    center = edge_index[0]
    counts = torch.bincount(center, minlength=N)
    rowptr = torch.zeros(N + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(counts, 0)
    e_atom = torch.zeros((N, 1), dtype=pos.dtype)
    forces = torch.zeros_like(pos)
    a0 = 0
    while a0 < N:
        a1 = int(torch.searchsorted(rowptr, rowptr[a0] + max_edges, right=True)) - 1
        a1 = max(a1, a0 + 1)
        a1 = min(a1, N)
        e0, e1 = int(rowptr[a0]), int(rowptr[a1])
        if e1 > e0:
            out = allegro_energy_forces(cfg, sd, pos, edge_index[:, e0:e1], atom_types,
                                        None if shift_vec is None else shift_vec[e0:e1])
            e_atom[a0:a1] = out["atomic_energy"][a0:a1]
            forces += out["forces"]
        a0 = a1
    return {"atomic_energy": e_atom, "total_energy": e_atom.sum().reshape(1, 1), "forces": forces}
