"""Whole-step TRAINING evaluation of `HipAllegroModel`.

The reference trains every MLP and every tensor-product path weight through autograd (weights are `Parameter`s:
allegro/nn/_allegro.py:192-213, allegro/nn/_strided/_contract.py:172-177) and a force-matching loss differentiates
the forces -- themselves a gradient -- again.  The inference pipeline of `aa_model_energy_forces` has hand-written
first derivatives with respect to positions only, so training mode evaluates the same function as a differentiable
graph instead:

* the strided tensor products -- the part of the model the reference's accelerators replace -- run on the HIP
  kernels through `ops.contract_segments_differentiable` (forward, both input gradients and the path-weight gradient
  on true center segments, closed under differentiation of any order);
* the per-edge linear layers are plain library GEMMs (`torch.matmul` = rocBLAS / hipBLASLt), the two-body embedding,
  spherical harmonics, weighted channels and the edge -> atom reduction are device-side torch ops: everything autograd
  already differentiates to any order.

No CPU path: tensors must live on the GPU (`_require_gpu`), the tensor products fail loudly without the HIP library.
"""
import math
from typing import Dict, List, Optional

import torch

from . import o3, ops

_ACT = {"silu": torch.nn.functional.silu, "mish": torch.nn.functional.mish, "gelu": torch.nn.functional.gelu, None: None}


def _real_sh(vec: torch.Tensor, l_max: int) -> torch.Tensor:
    """Component-normalised real spherical harmonics of the direction of `vec`, l = 0..l_max (<= 3), m = -l..l, y the polar
    axis -- the basis of allegro/nn/tensorembed.py:55-57,92 (same polynomials as csrc/aa_geom.h)."""
    if not 0 <= l_max <= 3:
        raise NotImplementedError("spherical harmonics: l_max <= 3")
    n = vec / vec.norm(dim=-1, keepdim=True)
    x, y, z = n.unbind(-1)
    cols = [torch.ones_like(x)]
    if l_max >= 1:
        c = math.sqrt(3.0)
        cols += [c * x, c * y, c * z]
    if l_max >= 2:
        a, b = math.sqrt(15.0), math.sqrt(5.0)
        xx, yy, zz = x * x, y * y, z * z
        cols += [a * x * z, a * x * y, b * (yy - 0.5 * (xx + zz)), a * y * z, 0.5 * a * (zz - xx)]
    if l_max >= 3:
        xx, yy, zz = x * x, y * y, z * z
        p, q, s, t = math.sqrt(70.0) / 4, math.sqrt(105.0), math.sqrt(42.0) / 4, math.sqrt(7.0) / 2
        cols += [p * x * (3 * zz - xx), q * x * y * z, s * x * (5 * yy - 1), t * y * (5 * yy - 3), s * z * (5 * yy - 1),
                 0.5 * q * y * (zz - xx), p * z * (zz - 3 * xx)]
    return torch.stack(cols, dim=-1)


def _mlp(x: torch.Tensor, weights: List[torch.Tensor], act: Optional[str], act_const: float, forward_init: bool, lib_id: Optional[int] = None) -> torch.Tensor:
    """nequip ScalarMLPFunction (EXT; call sites _allegro.py:90-94,193-213, tensorembed.py:76-81, allegro_models.py:173,231):
    bias-free linears scaled by 1/sqrt(fan_in) (fan_out if not forward_normalize), the activation between layers followed by
    its second-moment constant.  With `lib_id` the layers run through `ops.linear` (weight gradients on the hand-written
    slab-reduction kernel `aa_linear_wgrad`), otherwise through plain matmuls (the checker of tests/test_training.py)."""
    carry = 1.0
    last = len(weights) - 1
    for i, w in enumerate(weights):
        fan = w.shape[0] if forward_init else w.shape[1]
        ws = w * (carry / math.sqrt(float(fan)))
        x = ops.linear(x, ws, lib_id) if lib_id is not None else x @ ws
        if i < last and act is not None:
            x = ops.activation(x, act, lib_id) if (lib_id is not None and act in ops.ACT_CODES) else _ACT[act](x)
            carry = act_const
    return x


def _head_columns(x: torch.Tensor, a: int, b: int, lib_id: Optional[int] = None):
    """x[:, :a], x[:, a:a+b] as ONE autograd node (`split`: its backward is a single concatenation of the two gradients; two slices
    are two zero-filled [E, width] buffers, two copies and an addition -- and again in the second derivative)."""
    rest = x.shape[1] - a - b
    if b == 0 and rest == 0:
        return x, x[:, :0]
    sizes = [a, b] + ([rest] if rest else [])
    parts = torch.split(x, sizes, dim=1) if lib_id is None else ops.split_columns(x, sizes, lib_id)
    return parts[0], parts[1]


def _weighted_channels(sh: torch.Tensor, w: torch.Tensor, u: int, l_max: int) -> torch.Tensor:
    """MakeWeightedChannels (allegro/nn/_strided/_channels.py:44-63): out[e,c,i] = sh[e,i] * w[e,c,irrep(i)] (one weight per
    irrep) or sh[e,i] * w[e,c] (`weight_individual_irreps=False`)."""
    if w.shape[1] == u:
        return w.unsqueeze(-1) * sh.unsqueeze(1)
    # (broadcast views per irrep, not an index gather: the gather's backward is a sort-based index_put)
    wr = w.reshape(sh.shape[0], u, l_max + 1)
    return sh.unsqueeze(1) * torch.cat([wr[:, :, l:l + 1].expand(-1, -1, 2 * l + 1) for l in range(l_max + 1)], dim=-1)


class TrainingEvaluator:
    """Differentiable evaluation of one `HipAllegroModel`; owns the per-layer tensor-product plans (unregistered
    `HipContracter`s: the path weights stay the model's own parameters, state_dict keys unchanged)."""

    def __init__(self, model):
        from .nn import HipContracter, second_moment_const

        self.model = model
        hp = model.hparams
        env = o3.Irreps.spherical_harmonics(hp["l_max"], p=-1)
        self.sf = 1.0 / math.sqrt(hp["avg_num_neighbors"])
        self.contracters = []
        prev = torch.get_default_dtype()
        torch.set_default_dtype(model.dtype)
        try:
            for l in range(hp["num_layers"]):
                c = HipContracter(str(model.tps_irreps[l]), str(env), str(model.tps_irreps[l + 1]), mul=hp["num_tensor_features"],
                                  path_channel_coupling=hp["coupling"], scatter_factor=self.sf)
                ref = model._sd()[f"allegro.tps.{l}.w3j"]
                if c.w3j.shape != ref.shape or not torch.allclose(c.w3j.to(ref), ref):
                    raise RuntimeError(f"layer {l}: the model's w3j buffer is not the one its irreps generate -- cannot train it")
                if model._bound_lib is not None:
                    c._bind_library(model._bound_lib)
                self.contracters.append(c)
        finally:
            torch.set_default_dtype(prev)
        self.act_consts = {nl: second_moment_const(nl) for nl in set(model.nonlinearities) | {"silu"}}
        self._bessel_conv = None
        # hand-written training kernels around the tensor products (round 5: `aa_linear_wgrad`, `aa_weighted_channels`);
        # AA_TRAIN_EAGER=1 keeps the library-GEMM / eager-elementwise forms of round 3 (the checker and the A/B baseline)
        import os

        self.lib_id = None if os.environ.get("AA_TRAIN_EAGER", "0")[:1] == "1" else self.contracters[0]._lib_id

    # ------------------------------------------------------------------------------------------------------------
    def _weights(self, prefix: str) -> List[torch.Tensor]:
        node = self.model.func
        for part in prefix.split("."):
            node = getattr(node, part)
        out, i = [], 0
        while hasattr(node, str(i)):
            out.append(getattr(node, str(i)).weight)
            i += 1
        return out

    def _param(self, dotted: str) -> torch.Tensor:
        node = self.model.func
        for part in dotted.split("."):
            node = getattr(node, part)
        return node

    def _rows(self, table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """table[idx] for a small trainable table (type / class embeddings).  The gradient of an `index_select` is an atomic
        `index_add` of E rows onto a handful of table rows (one row for a single species: 1.9 ms per call at C3,
        profiles/r05_v11_train_c3_kernel_stats.txt); as a product with the one-hot matrix of `idx` the same gradient is the
        slab-reduced weight-gradient kernel -- deterministic, and 50x faster."""
        rows = table.shape[0]
        if self.lib_id is None or rows > 64:
            return table.index_select(0, idx)
        onehot = torch.nn.functional.one_hot(idx, rows).to(table.dtype)
        return ops.linear(onehot, table, self.lib_id)

    def _cat(self, xs):
        return torch.cat(xs, dim=-1) if self.lib_id is None else ops.cat_features(xs, self.lib_id)

    def _wc(self, sh, w, u, l_max):
        if self.lib_id is None:
            return _weighted_channels(sh, w, u, l_max)
        with _device_guard(sh):
            return ops.weighted_channels(sh, w, u, l_max, self.lib_id)

    def _two_body(self, x: torch.Tensor, tc: torch.Tensor, tn: torch.Tensor) -> torch.Tensor:
        m, hp = self.model, self.model.hparams
        T = len(m.type_names)
        if m.embed_kind == 1:
            # TwoBodySplineScalarEmbed (scalarembed.py:157-175) -> PerClassSpline.forward (spline.py:64-89)
            lower, upper = self._param("radial_chemical_embed.spline.lower"), self._param("radial_chemical_embed.spline.upper")
            k = 2 * math.pi / float(upper[0] - lower[0])
            t = k * (torch.minimum(torch.maximum(x, lower), upper) - lower)
            basis = 0.25 * (1 - torch.cos(t)).square()
            w = self._rows(self._param("radial_chemical_embed.spline.class_embed.weight"), tc * T + tn).view(x.shape[0], -1, lower.numel())
            return torch.bmm(w, basis.unsqueeze(-1)).squeeze(-1)
        # TwoBodyBesselScalarEmbed (scalarembed.py:60-81): Bessel x polynomial cutoff -> linear, times the type-pair embedding
        # (ProductTypeEmbedding, _edgeembed.py:68-84)
        bw = self._param("radial_chemical_embed.bessel_encode.bessel_weights")
        conv = m.bessel_convention if self._bessel_conv is None else self._bessel_conv
        if conv == "auto":  # (resolved once: the roots are a buffer of this model, and the test is a host read)
            roots = bw.reshape(-1).double()
            n = torch.arange(1, roots.numel() + 1, dtype=torch.float64, device=roots.device)
            if torch.allclose(roots, n, rtol=1e-5, atol=0):
                conv = "sinc"
            elif torch.allclose(roots, n * math.pi, rtol=1e-5, atol=0):
                conv = "npi"
            else:
                raise RuntimeError("Bessel roots are neither n nor n*pi: pass bessel_convention= to the model")
            self._bessel_conv = conv
        bessel = torch.sinc(x * bw) * bw if conv == "sinc" else torch.sin(bw * x) / x
        p = hp["poly_p"]
        cut = 1.0 - ((p + 1) * (p + 2) / 2) * x ** p + p * (p + 2) * x ** (p + 1) - (p * (p + 1) / 2) * x ** (p + 2)
        bessel = bessel * (cut * (x < 1.0))
        basis = _mlp(bessel, self._weights("radial_chemical_embed.type_embed.basis_linear.mlp"), "silu", self.act_consts["silu"],
                     hp["forward_normalize"], self.lib_id)
        # (index_select: its backward is an index_add, not the sort-based index_put of advanced indexing)
        pair = torch.cat((self._rows(self._param("radial_chemical_embed.type_embed.center_embed.weight"), tc),
                          self._rows(self._param("radial_chemical_embed.type_embed.neighbor_embed.weight"), tn)), dim=-1)
        return pair * basis

    def atomic_energy(self, pos: torch.Tensor, graph, shift_vec: Optional[torch.Tensor]) -> torch.Tensor:
        """[N,1] per-atom energies of the center-sorted `graph` (nn.PreparedGraph) as a differentiable function of `pos`,
        `shift_vec` and every parameter of the model (module order: allegro/model/allegro_models.py:222-228,262-268,297)."""
        m, hp = self.model, self.model.hparams
        S, u, L, l_max = hp["num_scalar_features"], hp["num_tensor_features"], hp["num_layers"], hp["l_max"]
        fwd = hp["forward_normalize"]
        center, nbr = graph.center.long(), graph.nbr.long()
        N = graph.num_atoms
        types = graph.types.long()
        # edge vectors, normalised lengths (tensorembed.py:86; allegro_models.py:153-157)
        hand = self.lib_id is not None and graph.t_rowptr is not None and graph.t_perm is not None
        if hand:
            with _device_guard(pos):
                vec = ops.edge_difference(pos, graph, self.lib_id)
        else:
            vec = pos.index_select(0, nbr) - pos.index_select(0, center)
        if shift_vec is not None:
            vec = vec + shift_vec
        tc, tn = types[center], types[nbr]
        recip = self._param("edge_norm.rmax_recip")
        x = (vec.norm(dim=-1) * (recip[tc, tn] if recip.numel() > 1 else recip.reshape(-1)[0])).unsqueeze(-1)
        emb = self._two_body(x, tc, tn)
        nl_embed, nl_latent, nl_readout = m.nonlinearities
        emb = _mlp(emb, self._weights("scalar_embed_mlp.mlp.mlp"), nl_embed, self.act_consts[nl_embed], fwd, self.lib_id)  # allegro_models.py:173-183
        silu_c = self.act_consts["silu"]
        # tensor embedding (tensorembed.py:85-96) and the first layer's environment weights (_allegro.py:251-258)
        sh = _real_sh(vec, l_max)
        tf = self._wc(sh, _mlp(emb, self._weights("tensor_embed.env_embed_linear.mlp"), "silu", silu_c, fwd, self.lib_id), u, l_max)
        We = (l_max + 1) * u if m.weight_individual_irreps else u
        proj = _mlp(emb, self._weights("allegro.first_layer_env_embed_projection.mlp"), "silu", silu_c, fwd, self.lib_id)
        first, env_w = _head_columns(proj, S, We, self.lib_id)
        scalars = [first]
        for l in range(L):  # _allegro.py:262-294
            c = self.contracters[l]
            env = self._wc(sh, env_w, u, l_max)
            with _device_guard(pos):
                tf = ops.contract_segments_differentiable(tf.reshape(-1, u, c.base_dim1), env, self._param(f"allegro.tps.{l}.weights"),
                                                          graph.rowptr, None, center, N, self.sf, c._plan(pos.dtype, pos.device),
                                                          c._lib_id, c.base_dim1, c.base_dim2, c.base_dim_out)
            if self.lib_id is None:
                tf_scalars = tf[:, :, 0]
            elif l < L - 1:
                tf, tf_scalars = ops.fork_scalars(tf, self.lib_id)  # (one node: tf goes on to the next layer, its scalars to the MLP)
            else:
                tf_scalars = tf.reshape(tf.shape[0], u) if tf.shape[2] == 1 else tf[:, :, 0]
            lat = _mlp(self._cat(scalars + [tf_scalars]), self._weights(f"allegro.latents.{l}.mlp"), nl_latent,
                       self.act_consts[nl_latent], fwd, self.lib_id)
            if l < L - 1:
                head, env_w = _head_columns(lat, S, We, self.lib_id)
            else:
                head, _ = _head_columns(lat, S, 0, self.lib_id)
            scalars.append(head)
        # edge readout, edge -> atom sum, per-type scale / shift (allegro_models.py:231-260; edgewise.py:40-60)
        e_edge = _mlp(self._cat(scalars), self._weights("edge_readout.mlp.mlp"), nl_readout, self.act_consts[nl_readout], fwd, self.lib_id)
        e_edge = e_edge * (1.0 / math.sqrt(2 * hp["avg_num_neighbors"]))
        if self.lib_id is not None:
            with _device_guard(pos):
                e_atom = ops.edge_to_atom_sum(e_edge, graph, self.lib_id)
        else:
            e_atom = torch.zeros((N, 1), dtype=e_edge.dtype, device=e_edge.device).index_add(0, center, e_edge)
        if m.has_scales:
            e_atom = e_atom * self._param("per_type_energy_scale_shift.scales").index_select(0, types).reshape(-1, 1)
        if m.has_shifts:
            e_atom = e_atom + self._param("per_type_energy_scale_shift.shifts").index_select(0, types).reshape(-1, 1)
        return e_atom

    def forward(self, data: Dict[str, torch.Tensor], graph) -> Dict[str, torch.Tensor]:
        """AtomicDataDict out with `forces` (and `stress` / `virial` when `cell` is given) that stay attached to the
        autograd graph (`create_graph=True`): a loss on them back-propagates into every parameter -- what the reference's
        ForceStressOutput wrapper does around its energy model (allegro_models.py:101-103)."""
        m = self.model
        from .nn import _require_gpu

        _require_gpu(m._get_lib(), data["pos"], "HipAllegroModel (training)")
        pos = data["pos"].to(m.dtype)
        if not pos.requires_grad:
            pos = pos.detach().requires_grad_(True)
        shift = graph.shift_vec
        wrt, eps = [pos], None
        pos_in = pos
        cells = None
        if "cell" in data:
            # strain-displacement construction (nequip ForceStressOutput, EXT): x -> x + x @ eps^T, differentiated at eps = 0
            cells = data["cell"].to(m.dtype).reshape(-1, 3, 3)
            nf = cells.shape[0]
            eps = torch.zeros(nf, 3, 3, dtype=m.dtype, device=pos.device, requires_grad=True)
            sym = 0.5 * (eps + eps.transpose(1, 2))
            frame_of_atom = data["batch"] if nf > 1 else torch.zeros(pos.shape[0], dtype=torch.long, device=pos.device)
            pos_in = pos + torch.einsum("ni,nji->nj", pos, sym[frame_of_atom])
            if shift is not None:
                shift = shift + torch.einsum("ei,eji->ej", shift, sym[frame_of_atom[graph.center.long()]])
            wrt.append(eps)
        e_atom = self.atomic_energy(pos_in, graph, shift)
        out = dict(data)
        out["atomic_energy"] = e_atom
        if "batch" in data:
            nf = int(cells.shape[0]) if cells is not None else int(data["batch"].max()) + 1
            total = torch.zeros(nf, 1, dtype=m.dtype, device=pos.device).index_add(0, data["batch"], e_atom)
        else:
            total = e_atom.sum().reshape(1, 1)
        out["total_energy"] = total
        grads = torch.autograd.grad(total.sum(), wrt, create_graph=torch.is_grad_enabled() and m.training)
        out["forces"] = -grads[0]
        if eps is not None:
            vol = torch.linalg.det(cells).abs().reshape(-1, 1, 1)
            out["stress"] = grads[1] / vol
            out["virial"] = -grads[1]
        return out


class ChunkedTrainingStep:
    """Exact gradient of a loss L(forces, total_energy) with activations of ONE atom block at a time (large boxes: the
    differentiable graph of a whole 10^5-atom frame needs ~1.5 KB of activations per edge and order of differentiation).

    By strict locality E_tot = sum_B E_B over blocks B of center atoms and F = -sum_B dE_B/dpos, so with the cotangents
    r_F = dL/dF, r_E = dL/dE_tot -- available after ONE evaluation of forces and energy, which the inference kernels deliver --

        dL/dtheta = sum_B d/dtheta [ r_E E_B - <r_F, dE_B/dpos> ]           (r_F, r_E held constant)

    Every term needs the graph of one block only (the block's edges; its atoms = the block + their neighbours, compact local
    numbering); the terms are back-propagated one after the other and accumulate in the parameters' `.grad`.  Same numbers as
    `loss_fn(forces, E).backward()` through the whole frame (tests/test_training.py), peak memory proportional to the block.
    The reference has no counterpart (it differentiates the whole batch at once; allegro/nn/_allegro.py is agnostic of it)."""

    def __init__(self, evaluator: "TrainingEvaluator", graph, max_edges_per_chunk: int):
        from .nn import PreparedGraph

        self.ev, self.graph = evaluator, graph
        rowptr = graph.rowptr.cpu().long()
        N, E = graph.num_atoms, graph.num_edges
        cuts, a = [0], 0
        while a < N:  # blocks of consecutive center atoms holding <= max_edges_per_chunk edges (at least one atom)
            target = int(rowptr[a]) + int(max_edges_per_chunk)
            b = int(torch.searchsorted(rowptr, torch.tensor(target), right=True)) - 1
            b = max(a + 1, min(b, N))
            cuts.append(b)
            a = b
        self.chunks = []
        center, nbr = graph.center.long(), graph.nbr.long()
        dev = center.device
        # atoms of edge-free blocks (isolated atoms: E_i = shift of their type) have no graph to back-propagate through, but their
        # share of dL/dshifts = r_E * (count per type) belongs to the gradient: counted here, added in `step`
        self.edge_free_types = torch.zeros(0, dtype=torch.long, device=dev)
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            e0, e1 = int(rowptr[a0]), int(rowptr[a1])
            if e1 == e0:
                self.edge_free_types = torch.cat([self.edge_free_types, graph.types.long()[a0:a1]])
                continue
            own = torch.arange(a0, a1, device=dev)
            nb = nbr[e0:e1]
            ghosts = torch.unique(nb[(nb < a0) | (nb >= a1)])
            local = torch.cat([own, ghosts])
            lookup = torch.full((N,), -1, dtype=torch.long, device=dev)
            lookup[local] = torch.arange(local.numel(), device=dev)
            ei = torch.stack([center[e0:e1] - a0, lookup[nb]])
            sv = None if graph.shift_vec is None else graph.shift_vec[e0:e1]
            g = PreparedGraph(ei, graph.types.long().index_select(0, local), int(local.numel()), sv)
            self.chunks.append((local, a1 - a0, g))

    def step(self, pos: torch.Tensor, loss_fn):
        """`loss_fn(forces [N,3], total_energy [1,1]) -> scalar`.  Accumulates dL/dtheta into the parameters' `.grad`; returns
        (loss, forces, total_energy), all detached."""
        m = self.ev.model
        with torch.no_grad():  # pass 1: the hand-written inference pipeline
            e_atom, forces = m.energy_forces(pos.detach().to(m.dtype), self.graph)
            e_atom, forces = e_atom.clone(), forces.clone()
        f_leaf = forces.detach().requires_grad_(True)
        e_leaf = e_atom.sum().reshape(1, 1).detach().requires_grad_(True)
        loss = loss_fn(f_leaf, e_leaf)
        r_f, r_e = torch.autograd.grad(loss, [f_leaf, e_leaf], allow_unused=True)
        r_f = torch.zeros_like(f_leaf) if r_f is None else r_f
        r_e = torch.zeros_like(e_leaf) if r_e is None else r_e
        for local, n_own, g in self.chunks:  # pass 2: one block's graph at a time
            p = pos.detach().to(m.dtype).index_select(0, local).requires_grad_(True)
            e_loc = self.ev.atomic_energy(p, g, g.shift_vec)
            e_blk = e_loc[:n_own].sum()
            (gp,) = torch.autograd.grad(e_blk, p, create_graph=True)
            s = r_e.reshape(()) * e_blk - (r_f.index_select(0, local) * gp).sum()
            s.backward()
        if m.has_shifts and self.edge_free_types.numel():
            sh = self.ev._param("per_type_energy_scale_shift.shifts")
            if sh.requires_grad:
                cnt = torch.bincount(self.edge_free_types, minlength=sh.shape[0]).to(sh.dtype) * r_e.reshape(()).to(sh.dtype)
                sh.grad = cnt.reshape(sh.shape) if sh.grad is None else sh.grad + cnt.reshape(sh.shape)
        return loss.detach(), forces, e_leaf.detach()


class _device_guard:
    """The HIP launches go to the device of `t` (no-op for the CPU emulation library of the tests)."""

    def __init__(self, t: torch.Tensor):
        self.ctx = torch.cuda.device(t.device) if t.is_cuda else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
