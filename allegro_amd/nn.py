"""Host-side mirror of the reference's operator / module interface for the hot path.

* `HipContracter`  -- same constructor, parameters (`weights`), persistent buffer (`w3j`) and
  `forward(x1, x2, idxs, scatter_dim_size)` contract as `allegro.nn._strided.Contracter`
  (allegro/nn/_strided/_contract.py:33-211), backed by the HIP tensor-product operator.
* `HipAllegroModel` -- same hyper-parameters and the same `state_dict` keys as
  `allegro.model.AllegroModel` (allegro/model/allegro_models.py:101-103,112-300), so
  `hip_model.load_state_dict(reference_model.state_dict())` is the drop-in step
  (the reference's own modifier swap relies on exactly that, _contract.py:277).  `forward(data)`
  returns `atomic_energy`, `total_energy`, `forces` computed by one C-ABI call
  (`aa_model_energy_forces`): forward and the hand-written reverse pass, all in HIP.

PyTorch is used for device memory, streams and parameter bookkeeping only.  There is no CPU
fallback: on a tensor that is not on a GPU these modules raise.
"""
import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops, o3

_TORCH2AA = {torch.float32: _lib.AA_F32, torch.float64: _lib.AA_F64}


ACT_KINDS = {"silu": 0, "mish": 1, "gelu": 2, None: 3}


def second_moment_const(nonlinearity: Optional[str] = "silu") -> float:
    """normalize2mom constant of an activation, 1/sqrt(E_{z~N(0,1)}[act(z)^2]), by quadrature (the constant nequip's
    ScalarMLPFunction folds into the layer after an activation); 1 for `None`."""
    if nonlinearity is None:
        return 1.0
    z = np.linspace(-12.0, 12.0, 240001)
    w = np.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    if nonlinearity == "silu":
        a = z / (1.0 + np.exp(-z))
    elif nonlinearity == "mish":
        a = z * np.tanh(np.logaddexp(0.0, z))
    elif nonlinearity == "gelu":
        from math import erf

        a = 0.5 * z * (1.0 + np.vectorize(erf)(z / math.sqrt(2.0)))
    else:
        raise NotImplementedError(f"nonlinearity {nonlinearity!r}: the reference offers silu, mish, gelu, None")
    f = a * a * w
    return 1.0 / math.sqrt(float(np.sum((f[1:] + f[:-1]) * 0.5 * (z[1] - z[0]))))


def silu_second_moment_const() -> float:
    return second_moment_const("silu")


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _device_ctx(device):
    """Make `device` the current HIP device for plan creation / packing / launches (plans and packed weights are
    device-specific; kernels are enqueued on that device's current stream)."""
    device = torch.device(device)
    if device.type == "cuda":
        return torch.cuda.device(device)
    import contextlib

    return contextlib.nullcontext()


def _require_gpu(lib: _lib.AllegroLib, t: torch.Tensor, what: str):
    if not t.is_cuda and not lib.is_emulation:
        raise _lib.AllegroError(f"{what}: tensors must live on the GPU (got {t.device}); there is no CPU fallback")


# ------------------------------------------------------------------------------------------------
# irreps logic of the Allegro layer stack
# ------------------------------------------------------------------------------------------------
def allegro_layer_irreps(l_max: int, parity: bool, num_layers: int) -> List[o3.Irreps]:
    """Per-layer tensor-track irreps: forward reachability then backward pruning.
    Mirrors allegro/nn/_allegro.py:101-160 (and the allowed set of allegro_models.py:76-86).
    Returns tps_irreps[0..L]: input irreps of layer l = [l], output = [l+1]."""
    env = o3.Irreps.spherical_harmonics(l_max, p=-1)
    if parity:
        allowed = o3.Irreps([(1, (l, p)) for l in range(l_max + 1) for p in (1, -1)])
    else:
        allowed = env
    arg = env
    tps = [arg]
    for layer in range(num_layers):
        ir_out = o3.Irreps([(1, (0, 1))]) if layer == num_layers - 1 else allowed
        ir_out = o3.Irreps([(m, ir) for m, ir in ir_out if o3.tp_path_exists(arg, env, ir)])
        arg = ir_out
        tps.append(ir_out)
    out = tps[-1]
    new = [out]
    for arg in reversed(tps[:-1]):
        keep = []
        for m, arg_ir in arg:
            if any(any(i in out for i in arg_ir * env_ir) for _, env_ir in env):
                keep.append((m, arg_ir))
        out = o3.Irreps(keep)
        new.append(out)
    tps = list(reversed(new))
    assert tps[-1].lmax == 0
    return tps


def build_w3j(irreps_in1: o3.Irreps, irreps_in2: o3.Irreps, irreps_out: o3.Irreps,
              instructions: Optional[Sequence[Tuple[int, int, int]]] = None, irrep_normalization="component"):
    """The dense `w3j` buffer exactly as Contracter.__init__ lays it out (_contract.py:48-168):
    instructions enumerated in (i_out, i_1, i_2) loop order, values x sqrt(2 l_out + 1), [p,i,k] when
    ij-diagonal else [p,i,j,k], path dim squeezed when there is a single path."""
    if instructions is None:
        instructions = []
        for i_out, (_, ir_out) in enumerate(irreps_out):
            for i_1, (_, ir_1) in enumerate(irreps_in1):
                for i_2, (_, ir_2) in enumerate(irreps_in2):
                    if ir_out in (ir_1 * ir_2):
                        instructions.append((i_1, i_2, i_out))
    assert len(instructions) > 0, "No TP paths available"
    o1, o2, oo = irreps_in1.offsets(), irreps_in2.offsets(), irreps_out.offsets()
    d1, d2, dout = irreps_in1.dim, irreps_in2.dim, irreps_out.dim
    blocks = []
    for (i_1, i_2, i_out) in instructions:
        ir1, ir2, iro = irreps_in1[i_1][1], irreps_in2[i_2][1], irreps_out[i_out][1]
        assert ir1.p * ir2.p == iro.p and abs(ir1.l - ir2.l) <= iro.l <= ir1.l + ir2.l
        w = o3.wigner_3j(ir1.l, ir2.l, iro.l)
        if irrep_normalization == "component":
            w = w * math.sqrt(2 * iro.l + 1)
        elif irrep_normalization is not None:
            raise NotImplementedError(irrep_normalization)
        blocks.append((o1[i_1], o2[i_2], oo[i_out], w))
    diag = d1 == d2 and all(
        all(o_a + i == o_b + j for i, j, _ in zip(*np.nonzero(w))) for o_a, o_b, _, w in blocks)
    P = len(instructions)
    if diag:
        w3j = np.zeros((P, d1, dout))
        for p, (a, b, c, w) in enumerate(blocks):
            for i, j, k in zip(*np.nonzero(w)):
                w3j[p, a + i, c + k] = w[i, j, k]
    else:
        w3j = np.zeros((P, d1, d2, dout))
        for p, (a, b, c, w) in enumerate(blocks):
            w3j[p, a:a + w.shape[0], b:b + w.shape[1], c:c + w.shape[2]] = w
    if P == 1:
        w3j = w3j[0]
    return w3j, list(instructions), bool(diag), (d1, d2, dout)


def w3j_to_desc(w3j: torch.Tensor, mul: int, dims: Tuple[int, int, int], num_paths: int, diag: bool, coupling: bool):
    """Sparse non-zeros of a (possibly squeezed) w3j buffer -> C descriptor."""
    d1, d2, dout = dims
    w = w3j.detach().to("cpu", torch.float64).numpy()
    if num_paths == 1:
        w = w[None]
    if diag:
        p, i, k = np.nonzero(w)
        j = i
        val = w[p, i, k]
    else:
        p, i, j, k = np.nonzero(w)
        val = w[p, i, j, k]
    return _lib.make_tp_desc(mul, d1, d2, dout, num_paths, coupling, i, j, k, p, val)


# ------------------------------------------------------------------------------------------------
# segment bookkeeping
# ------------------------------------------------------------------------------------------------
def segments_from_index(idxs: torch.Tensor, num_segments: int, assume_sorted: bool = False):
    """rowptr[int32, N+1] and (if idxs is not sorted) the stable sort permutation eids[int32, E].
    `assume_sorted=True` skips the sortedness check, i.e. the only device->host synchronisation of this function
    (the caller vouches that `idxs` is non-decreasing, as the center index of a model's edge list is)."""
    idxs = idxs.reshape(-1)
    is_sorted = True
    if not assume_sorted and idxs.numel() > 1:
        is_sorted = bool((idxs[1:] >= idxs[:-1]).all())
    eids = None
    if not is_sorted:
        eids = torch.argsort(idxs, stable=True).to(torch.int32)
    counts = torch.bincount(idxs, minlength=num_segments)
    rowptr = torch.zeros(num_segments + 1, dtype=torch.int32, device=idxs.device)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return rowptr, eids


class _SegmentCache:
    """rowptr / eids of the most recent index tensors, keyed on (storage address, in-place version, length, segment
    count).  The keyed tensor is kept alive by the entry, so its address cannot be recycled for another tensor
    while the entry exists.  All Contracters of a model receive the SAME `idxs` tensor in every layer
    (allegro/nn/_allegro.py:238,268), so one entry serves the whole forward; a handful covers train/val alternation."""

    def __init__(self, capacity: int = 4):
        self.capacity = capacity
        self.entries = []  # [(key, idxs_ref, rowptr, eids)]

    def get(self, idxs: torch.Tensor, num_segments: int, assume_sorted: bool):
        key = (idxs.data_ptr(), int(idxs._version), int(idxs.numel()), int(num_segments), str(idxs.device), bool(assume_sorted))
        for i, (k, _ref, rowptr, eids) in enumerate(self.entries):
            if k == key:
                if i:
                    self.entries.insert(0, self.entries.pop(i))
                return rowptr, eids
        rowptr, eids = segments_from_index(idxs, num_segments, assume_sorted)
        self.entries.insert(0, (key, idxs, rowptr, eids))
        del self.entries[self.capacity:]
        return rowptr, eids


_SEGMENTS = _SegmentCache()


def _is_tracing() -> bool:
    """True under torch.compile / torch.export tracing (FakeTensors: no addresses, no values)."""
    c = torch.compiler
    return bool(c.is_compiling() or (hasattr(c, "is_exporting") and c.is_exporting()))


# ------------------------------------------------------------------------------------------------
# Contracter (seam B1/B2)
# ------------------------------------------------------------------------------------------------
class HipContracter(torch.nn.Module):
    """Drop-in for `Contracter` (allegro/nn/_strided/_contract.py:11-251) on the HIP operator."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, mul: int, instructions=None,
                 path_channel_coupling: bool = True, scatter_factor: Optional[float] = None,
                 irrep_normalization: str = "component"):
        super().__init__()
        self.irreps_in1, self.irreps_in2, self.irreps_out = o3.Irreps(irreps_in1), o3.Irreps(irreps_in2), o3.Irreps(irreps_out)
        for irr in (self.irreps_in1, self.irreps_in2, self.irreps_out):
            assert all(m == 1 for m, _ in irr)
        assert mul > 0
        self.mul = mul
        self.scatter_factor = scatter_factor
        self.irrep_normalization = irrep_normalization
        self.path_channel_coupling = path_channel_coupling
        w3j, instr, diag, (d1, d2, dout) = build_w3j(self.irreps_in1, self.irreps_in2, self.irreps_out, instructions,
                                                      irrep_normalization)
        self.instructions = instructions
        self.num_paths = len(instr)
        self.w3j_is_ij_diagonal = diag
        self.base_dim1, self.base_dim2, self.base_dim_out = d1, d2, dout
        self.register_buffer("w3j", torch.tensor(w3j, dtype=torch.get_default_dtype()))
        shape = (mul,) if path_channel_coupling else tuple()
        if self.num_paths > 1:
            shape = shape + (self.num_paths,)
        self.weights = torch.nn.Parameter(torch.empty(shape).uniform_(-math.sqrt(3), math.sqrt(3)))
        self._plans: Dict[tuple, int] = {}  # (dtype, device index) -> aa_tp_plan*
        self._keep = []
        self._bound_lib: Optional[_lib.AllegroLib] = None
        self._lib_id = 0
        # set True when the caller guarantees center-sorted `idxs` (skips the sortedness check and its host sync)
        self.assume_sorted_idxs = False

    def _get_lib(self) -> _lib.AllegroLib:
        return self._bound_lib if self._bound_lib is not None else _lib.load()

    def _bind_library(self, lib: _lib.AllegroLib):
        """(tests) bind an explicitly loaded library instead of the default gfx950 one."""
        self._bound_lib = lib
        self._lib_id = ops.register_library(lib)
        self._plans.clear()

    def _plan(self, dtype, device=None) -> int:
        """The plan owns device-resident Clebsch-Gordan tables: one per (dtype, device), created with that device current."""
        dev = torch.device("cpu") if device is None else torch.device(device)
        key = (dtype, dev.index if dev.type == "cuda" else -1)
        if key not in self._plans:
            desc, keep = w3j_to_desc(self.w3j, self.mul, (self.base_dim1, self.base_dim2, self.base_dim_out),
                                     self.num_paths, self.w3j_is_ij_diagonal, self.path_channel_coupling)
            self._keep.append(keep)
            with _device_ctx(dev):
                self._plans[key] = self._get_lib().tp_plan_create(desc, _TORCH2AA[dtype])
        return self._plans[key]

    def forward(self, x1, x2, idxs, scatter_dim_size):
        if isinstance(scatter_dim_size, torch.Tensor):
            scatter_dim_size = int(scatter_dim_size.reshape(-1)[0])
        x1 = x1.reshape(-1, self.mul, self.base_dim1)
        x2 = x2.reshape(-1, self.mul, self.base_dim2)
        sf = 1.0 if self.scatter_factor is None else float(self.scatter_factor)
        if self.training and torch.is_grad_enabled():
            # training: the arbitrarily differentiable form of the WHOLE forward (scale + scatter + gather + contraction) on
            # the segmented kernels (ops.contract_segments_differentiable) -- a force-matching loss differentiates the
            # forces again.  (The reference's Triton contracter falls back to the eager formulation when training,
            # _flashallegro.py:725-755; its cuEquivariance contracter keeps the fused gather, _cueq_contracter.py:84-131.)
            if os.environ.get("AA_TRAIN_PER_EDGE", "0")[:1] == "1":  # (A/B: the first formulation, every edge its own segment)
                x2s = torch.zeros((int(scatter_dim_size),) + tuple(x2.shape[1:]), dtype=x2.dtype, device=x2.device)
                x2s = x2s.index_add(0, idxs.reshape(-1), sf * x2)
                return self._contract(x1, x2s.index_select(0, idxs.reshape(-1)))
            rowptr, eids = _SEGMENTS.get(idxs, scatter_dim_size, self.assume_sorted_idxs)
            _require_gpu(self._get_lib(), x1, "HipContracter")
            with _device_ctx(x1.device):
                return ops.contract_segments_differentiable(x1, x2, self.weights, rowptr, eids, idxs, int(scatter_dim_size), sf,
                                                            self._plan(x1.dtype, x1.device), self._lib_id, self.base_dim1,
                                                            self.base_dim2, self.base_dim_out)
        if _is_tracing():
            # torch.export / torch.compile: no data pointers, no data-dependent Python -- the bookkeeping is one opaque
            # op with a fake kernel (ops.segments); plan handles are process-local, so a traced program is valid in
            # THIS process (a Python-free host loads the whole-step op of allegro_amd/export.py instead)
            rowptr, eids = torch.ops.allegro_amd.segments(idxs, scatter_dim_size, self.assume_sorted_idxs)
            return self._op(x1, x2, rowptr, None if self.assume_sorted_idxs else eids, scatter_dim_size, sf)
        # per-call host sync / bincount / cumsum only on the first layer of the first forward with this index tensor
        rowptr, eids = _SEGMENTS.get(idxs, scatter_dim_size, self.assume_sorted_idxs)
        return self._op(x1, x2, rowptr, eids, int(scatter_dim_size), sf)

    def _op(self, x1, x2, rowptr, eids, num_atoms: int, scatter_factor: float):
        """The registered library op (allegro_amd/ops.py).  Autograd gives the x1 / x2 gradients and -- when the path
        weights require grad (training, like the reference's eager and cuEquivariance contracters,
        _contract.py:172-177) -- the weight gradient from `aa_tp_backward_weights`."""
        _require_gpu(self._get_lib(), x1, "HipContracter")
        with _device_ctx(x1.device):
            out, _x2s = torch.ops.allegro_amd.tp_forward(x1, x2, self.weights, rowptr, eids, num_atoms,
                                                         scatter_factor, self._plan(x1.dtype, x1.device), self._lib_id,
                                                         self.base_dim2, self.base_dim_out)
        return out

    def _contract(self, x1, x2):
        """Contraction only (seam B1, _contract.py:213): every edge is its own segment.  Differentiable to any order
        (x1, x2 and the path weights)."""
        _require_gpu(self._get_lib(), x1, "HipContracter")
        with _device_ctx(x1.device):
            return ops.contract_differentiable(x1, x2, self.weights, self._plan(x1.dtype, x1.device), self._lib_id,
                                               self.base_dim1, self.base_dim2, self.base_dim_out)

    @classmethod
    def from_contracter(cls, old: torch.nn.Module) -> "HipContracter":
        """Build from a reference `Contracter` instance (same constructor attributes and state_dict keys,
        _contract.py:72-76,165-177) -- the `factory` of the reference's `enable_*Contracter` modifiers (:262-279)."""
        prev = torch.get_default_dtype()
        torch.set_default_dtype(old.w3j.dtype)
        try:
            new = cls(irreps_in1=str(old.irreps_in1), irreps_in2=str(old.irreps_in2), irreps_out=str(old.irreps_out),
                      mul=old.mul, instructions=old.instructions, path_channel_coupling=old.path_channel_coupling,
                      scatter_factor=old.scatter_factor, irrep_normalization=old.irrep_normalization)
        finally:
            torch.set_default_dtype(prev)
        new.load_state_dict(old.state_dict())
        return new.to(old.weights.device)

    def extra_repr(self):
        return f"{self.irreps_in1} x {self.irreps_in2} -> {self.irreps_out} | {self.mul} channels | {self.num_paths} paths"

    def __del__(self):
        try:
            for h in self._plans.values():
                self._get_lib().tp_plan_destroy(h)
        except Exception:
            pass


def replace_submodules(model: torch.nn.Module, target_cls, factory) -> torch.nn.Module:
    """Recursively replace every instance of `target_cls` by `factory(old)` (nequip.nn.replace_submodules, EXT)."""
    if isinstance(model, target_cls):
        return factory(model)
    for name, child in list(model.named_children()):
        setattr(model, name, replace_submodules(child, target_cls, factory))
    return model


class _DuckContracterMeta(type):
    """isinstance(obj, AnyContracter) <=> obj carries a reference Contracter's attributes and is not already a
    HipContracter (so the modifier works on the reference's class without importing it)."""
    _ATTRS = ("irreps_in1", "irreps_in2", "irreps_out", "mul", "w3j", "weights", "path_channel_coupling")

    def __instancecheck__(cls, obj):
        return (not isinstance(obj, HipContracter)) and all(hasattr(obj, a) for a in cls._ATTRS)


class AnyContracter(metaclass=_DuckContracterMeta):
    pass


def enable_HipContracter(model: torch.nn.Module, contracter_cls=None) -> torch.nn.Module:
    """Model modifier: swap every `Contracter` of `model` for the HIP operator -- the analogue of
    `Contracter.enable_TritonContracter` (_contract.py:253-282).  `contracter_cls` defaults to any module that
    carries a Contracter's attributes.  Unlike the Triton kernel there is no build-time guard: ij-diagonal,
    single-path and uncoupled modes are all covered, and so is training mode (weight gradients).

    Discovery through nequip (`nequip.model.modify` in a config, `nequip-compile --modifiers enable_HipContracter`):
    `allegro_amd._nequip_ext.register()` -- run by the `nequip.extension` entry point of this package's
    pyproject.toml, like the reference's own `init_always = "allegro"` (pyproject.toml:50-51) -- attaches this
    function to the reference's `Contracter` as a `@model_modifier(persistent=False) @classmethod`, the exact form of
    `_contract.py:253-255,284-286`."""
    return replace_submodules(model, AnyContracter if contracter_cls is None else contracter_cls,
                              HipContracter.from_contracter)


def _enable_HipContracter_classmethod(cls, model):
    """Body of the classmethod attached to the reference's Contracter (and to HipContracter below): replaces
    instances of `cls` -- the signature nequip's modifier machinery calls, `_contract.py:255,282`."""
    target = AnyContracter if cls is HipContracter else cls
    return replace_submodules(model, target, HipContracter.from_contracter)


HipContracter.enable_HipContracter = classmethod(_enable_HipContracter_classmethod)


# ------------------------------------------------------------------------------------------------
# whole model
# ------------------------------------------------------------------------------------------------
class _Node(torch.nn.Module):
    """Plain container used to reproduce the reference's state_dict key hierarchy."""


def _ensure_path(root: torch.nn.Module, dotted: str) -> Tuple[torch.nn.Module, str]:
    parts = dotted.split(".")
    node = root
    for name in parts[:-1]:
        if not hasattr(node, name):
            node.add_module(name, _Node())
        node = getattr(node, name)
    return node, parts[-1]


_BESSEL_CONVENTIONS = {"auto": 0, "npi": 1, "sinc": 2}


class PreparedGraph:
    """Device-resident center-sorted CSR view of an edge list (the `aa_graph` struct)."""

    def __init__(self, edge_index: torch.Tensor, atom_types: torch.Tensor, num_atoms: int,
                 shift_vec: Optional[torch.Tensor] = None, transposed: bool = True,
                 rowptr: Optional[torch.Tensor] = None, lib: Optional[_lib.AllegroLib] = None):
        center = edge_index[0]
        self.perm = None
        # (a list that comes with its row pointers -- the device neighbour list -- is center-sorted by construction: no check, no sync)
        if rowptr is None and center.numel() > 1 and not bool((center[1:] >= center[:-1]).all()):
            self.perm = torch.argsort(center, stable=True)
            edge_index = edge_index[:, self.perm]
            if shift_vec is not None:
                shift_vec = shift_vec[self.perm]
        self.num_atoms, self.num_edges = int(num_atoms), int(edge_index.shape[1])
        self.center = edge_index[0].to(torch.int32).contiguous()
        self.nbr = edge_index[1].to(torch.int32).contiguous()
        if rowptr is not None:  # already known (device neighbor list)
            self.rowptr = rowptr.to(torch.int32).contiguous()
        else:
            counts = torch.bincount(edge_index[0], minlength=num_atoms)
            self.rowptr = torch.zeros(num_atoms + 1, dtype=torch.int32, device=edge_index.device)
            self.rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        self.types = atom_types.reshape(-1).to(torch.int32).contiguous()
        self.shift_vec = None if shift_vec is None else shift_vec.contiguous()
        # block of center atoms that have edges (the owned block of an atom-block partition, allegro_amd/dist.py):
        # the per-atom kernels only visit it.  One host read at graph-preparation time, none per step.
        self.atom_begin = self.atom_end = self.max_degree = 0
        # transposed CSR (edges grouped by neighbor): lets the library gather forces per atom in a fixed order
        # (bit-reproducible); without it neighbor contributions are accumulated with floating-point atomics
        self.t_perm = self.t_rowptr = None
        if lib is None and edge_index.is_cuda:
            lib = _lib.load()
        if transposed and lib is not None and (edge_index.is_cuda or lib.is_emulation):
            # one library call per neighbour list (`aa_graph_transpose`: counting sort + per-atom group sort + the three hints) and
            # ONE host read, instead of argsort / bincount / cumsum / max through a dozen tensor operations: an MD loop prepares a
            # graph per list (0.28 -> ~0.1 ms; profiles/r05_*_md_loop_c3.json)
            dev = edge_index.device
            self.t_rowptr = torch.empty(num_atoms + 1, dtype=torch.int32, device=dev)
            self.t_perm = torch.empty(self.num_edges, dtype=torch.int32, device=dev)
            hints = torch.empty(3, dtype=torch.int32, device=dev)
            nbytes = lib.lib.aa_graph_transpose_workspace_bytes(num_atoms)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            with _device_ctx(dev):
                lib.check(lib.lib.aa_graph_transpose(num_atoms, self.num_edges, self.rowptr.data_ptr(), self.nbr.data_ptr(), self.t_rowptr.data_ptr(),
                                                     self.t_perm.data_ptr() if self.num_edges else None, hints.data_ptr(), ws.data_ptr(), nbytes,
                                                     _stream_ptr(edge_index)), "aa_graph_transpose")
            self.atom_begin, self.atom_end, self.max_degree = (int(v) for v in hints.tolist())
            return
        if self.num_edges > 0:
            # largest edge segment: <= 32 selects the fused per-atom-tile kernels (one wave = one atom's MFMA tile); ONE host read
            # for the three hints
            deg_max = (self.rowptr[1:] - self.rowptr[:-1]).max().to(torch.int32)
            first, last, dmax = torch.stack([self.center[0], self.center[-1], deg_max]).tolist()
            self.atom_begin, self.atom_end, self.max_degree = int(first), int(last) + 1, int(dmax)
        if transposed:
            self.t_perm = torch.argsort(edge_index[1], stable=True).to(torch.int32).contiguous()
            self.t_rowptr = torch.zeros(num_atoms + 1, dtype=torch.int32, device=edge_index.device)
            self.t_rowptr[1:] = torch.cumsum(torch.bincount(edge_index[1], minlength=num_atoms), 0).to(torch.int32)

    def c_struct(self) -> _lib.Graph:
        return _lib.Graph(self.num_atoms, self.num_edges, self.center.data_ptr(), self.nbr.data_ptr(),
                          self.rowptr.data_ptr(), self.types.data_ptr(),
                          self.shift_vec.data_ptr() if self.shift_vec is not None else None,
                          self.t_rowptr.data_ptr() if self.t_rowptr is not None else None,
                          self.t_perm.data_ptr() if self.t_perm is not None else None,
                          self.atom_begin, self.atom_end, self.max_degree)


class DeviceNeighborList:
    """Result of `neighbor_list`: center-sorted edges on the device (`aa_nl_count` / `aa_nl_fill`)."""

    def __init__(self, edge_index, rowptr, cell_shift, shift_vec, lib=None):
        self.edge_index, self.rowptr, self.cell_shift, self.shift_vec, self._lib = edge_index, rowptr, cell_shift, shift_vec, lib

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def prepare(self, atom_types: torch.Tensor, transposed: bool = True) -> PreparedGraph:
        return PreparedGraph(self.edge_index, atom_types, self.rowptr.numel() - 1, self.shift_vec, transposed=transposed,
                             rowptr=self.rowptr, lib=self._lib)

    def ghost_layout(self, pos: torch.Tensor, atom_types: torch.Tensor):
        """The same list in the ghost-atom layout of the reference's `pair_allegro` contract (allegro/_compile.py:28-63),
        built on the device: every outside-cell edge gets its own ghost atom (not deduplicated) at `pos[nbr] + shift`,
        ghosts are appended after the real atoms, no cell / shifts remain; edges keep their center-sorted order.
        Returns (pos [N + G, 3], atom_types [N + G], edge_index int64 [2, E], ghost_source int64 [G]) -- forces on ghost
        row g belong to atom ghost_source[g] (what LAMMPS reverse-communicates).  One host read: the ghost count."""
        n = pos.shape[0]
        ei = self.edge_index.long()
        outside = (self.cell_shift != 0).any(dim=1)
        slot = torch.cumsum(outside.to(torch.int64), 0) - 1  # running ghost index of each outside-cell edge
        ghost_src = ei[1][outside]
        nbr = torch.where(outside, n + slot, ei[1])
        pos_ext = torch.cat((pos, pos.index_select(0, ghost_src) + self.shift_vec[outside].to(pos.dtype)), dim=0)
        types_ext = torch.cat((atom_types, atom_types.index_select(0, ghost_src)), dim=0)
        return pos_ext, types_ext, torch.stack((ei[0], nbr)), ghost_src


def neighbor_list(pos: torch.Tensor, cell, pbc, r_cut: float, lib: Optional[_lib.AllegroLib] = None) -> DeviceNeighborList:
    """Cell-list neighbor list on the device the positions live on: every pair/image with |r| < r_cut, edges sorted by
    center (`r_e = pos[nbr] - pos[center] + cell_shift @ cell`, the `with_edge_vectors_` convention).  Only the edge
    count crosses to the host (the outputs must be allocated).  `cell`: 3x3, lattice vectors as rows; `pbc`: 3 bools."""
    lib = lib if lib is not None else _lib.load()
    _require_gpu(lib, pos, "neighbor_list")
    assert pos.dtype in _TORCH2AA and pos.dim() == 2 and pos.shape[1] == 3
    pos = pos.detach().contiguous()
    N, dev = pos.shape[0], pos.device
    inp = _lib.NlInput()
    inp.num_atoms, inp.pos, inp.dtype, inp.r_cut = N, pos.data_ptr(), _TORCH2AA[pos.dtype], float(r_cut)
    cell_h = torch.as_tensor(cell, dtype=torch.float64).reshape(9).tolist()
    for q in range(9):
        inp.cell[q] = cell_h[q]
    if isinstance(pbc, bool):
        pbc = (pbc,) * 3
    for q in range(3):
        inp.pbc[q] = int(bool(pbc[q]))
    nbytes = lib.lib.aa_nl_workspace_bytes(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    n_edges = C.c_int64()
    stream = _stream_ptr(pos)
    lib.check(lib.lib.aa_nl_count(C.byref(inp), ws.data_ptr(), nbytes, rowptr.data_ptr(), C.byref(n_edges), stream),
              "aa_nl_count")
    E = int(n_edges.value)
    edge_index = torch.empty((2, E), dtype=torch.int32, device=dev)
    cell_shift = torch.empty((E, 3), dtype=torch.int32, device=dev)
    shift_vec = torch.empty((E, 3), dtype=pos.dtype, device=dev)
    if E > 0:
        lib.check(lib.lib.aa_nl_fill(C.byref(inp), ws.data_ptr(), nbytes, rowptr.data_ptr(), edge_index[0].data_ptr(),
                                     edge_index[1].data_ptr(), cell_shift.data_ptr(), shift_vec.data_ptr(), stream),
                  "aa_nl_fill")
    return DeviceNeighborList(edge_index, rowptr, cell_shift, shift_vec, lib)


class HipAllegroModel(torch.nn.Module):
    """Energy + forces of an Allegro model through the HIP hot path.

    Constructor arguments follow `allegro.model.AllegroModel` (allegro_models.py:112-147); only the
    Bessel two-body embedding (`allegro.nn.TwoBodyBesselScalarEmbed`) and SiLU MLPs are supported.
    """

    def __init__(self, *, type_names: Sequence[str], r_max: float, l_max: int, num_layers: int = 2,
                 num_scalar_features: int = 64, num_tensor_features: int = 16, parity: bool = True,
                 radial_chemical_embed: Optional[dict] = None, radial_chemical_embed_dim: Optional[int] = None,
                 per_edge_type_cutoff: Optional[dict] = None,
                 scalar_embed_mlp_hidden_layers_depth: int = 1, scalar_embed_mlp_hidden_layers_width: int = 64,
                 scalar_embed_mlp_nonlinearity: Optional[str] = "silu",
                 allegro_mlp_hidden_layers_depth: int = 1, allegro_mlp_hidden_layers_width: int = 64,
                 allegro_mlp_nonlinearity: Optional[str] = "silu", tp_path_channel_coupling: bool = True,
                 readout_mlp_hidden_layers_depth: int = 1, readout_mlp_hidden_layers_width: int = 32,
                 readout_mlp_nonlinearity: Optional[str] = "silu", avg_num_neighbors: Optional[float] = None,
                 weight_individual_irreps: bool = True, per_type_energy_scales=None, per_type_energy_shifts=None,
                 per_type_energy_scales_trainable: bool = False, per_type_energy_shifts_trainable: bool = False,
                 pair_potential=None, forward_normalize: bool = True, seed: Optional[int] = None,
                 model_dtype: str = "float32", compile_mode: Optional[str] = None,
                 bessel_convention: str = "auto"):
        super().__init__()
        assert avg_num_neighbors is not None, "`avg_num_neighbors` must be set for Allegro models"
        # which published form of nequip's BesselEdgeLengthEncoding (EXT) the state_dict follows: "npi" (roots n*pi,
        # sin(w x)/x), "sinc" (roots n, sinc(x w) w), or "auto" = told apart by the stored values -- which is only
        # possible for untrained roots (aa_model_config.bessel_convention)
        if bessel_convention not in _BESSEL_CONVENTIONS:
            raise ValueError(f"bessel_convention {bessel_convention!r}: one of {sorted(_BESSEL_CONVENTIONS)}")
        self.bessel_convention = bessel_convention
        if pair_potential is not None:
            raise NotImplementedError("pair potentials (nequip's ZBL module, EXT) are outside the hot path (DESIGN.md section 8)")
        self.nonlinearities = (scalar_embed_mlp_nonlinearity, allegro_mlp_nonlinearity, readout_mlp_nonlinearity)
        for nl in self.nonlinearities:
            if nl not in ACT_KINDS:
                raise NotImplementedError(f"nonlinearity {nl!r}: the reference offers silu, mish, gelu, None")
        self.weight_individual_irreps = bool(weight_individual_irreps)
        rce = dict(radial_chemical_embed or {})
        tgt = rce.pop("_target_", "allegro.nn.TwoBodyBesselScalarEmbed")
        self.embed_kind = {"TwoBodyBesselScalarEmbed": 0, "TwoBodySplineScalarEmbed": 1}.get(tgt.rsplit(".", 1)[-1])
        if self.embed_kind is None:
            raise NotImplementedError(f"radial_chemical_embed {tgt}: only allegro.nn.TwoBodyBesselScalarEmbed and "
                                      "allegro.nn.TwoBodySplineScalarEmbed exist in the reference (scalarembed.py)")
        if rce.get("bessel_trainable", False) and bessel_convention == "auto":
            raise NotImplementedError("trained Bessel roots cannot be told apart by their values: pass bessel_convention='sinc' "
                                      "or 'npi'")
        self.dtype = {"float32": torch.float32, "float64": torch.float64}[model_dtype]
        self.type_names = list(type_names)
        T = len(self.type_names)
        S, u, L = num_scalar_features, num_tensor_features, num_layers
        S0 = S if radial_chemical_embed_dim is None else radial_chemical_embed_dim
        # number of radial basis functions: Bessel roots (scalarembed.py:22) or splines (:108)
        B = int(rce.get("num_splines", 16)) if self.embed_kind == 1 else int(rce.get("num_bessels", 8))
        span = int(rce.get("spline_span", 12))
        assert self.embed_kind == 0 or 0 <= span <= B, "spline.py:32"
        self.hparams = dict(r_max=float(r_max), l_max=l_max, num_layers=L, num_scalar_features=S, num_tensor_features=u,
                            embed_dim=S0, num_bessels=B, poly_p=float(rce.get("polynomial_cutoff_p", 6)),
                            spline_span=span,
                            embed_depth=scalar_embed_mlp_hidden_layers_depth, embed_width=scalar_embed_mlp_hidden_layers_width,
                            latent_depth=allegro_mlp_hidden_layers_depth, latent_width=allegro_mlp_hidden_layers_width,
                            readout_depth=readout_mlp_hidden_layers_depth, readout_width=readout_mlp_hidden_layers_width,
                            coupling=bool(tp_path_channel_coupling), avg_num_neighbors=float(avg_num_neighbors),
                            forward_normalize=bool(forward_normalize), parity=bool(parity))
        if seed is not None:
            torch.manual_seed(seed)
        dt = self.dtype
        self.func = _Node()
        rng_u = lambda *shape: torch.empty(*shape, dtype=dt).uniform_(-math.sqrt(3), math.sqrt(3))  # noqa: E731

        def add(key, tensor, kind):
            node, leaf = _ensure_path(self.func, key)
            if kind == "param":
                node.register_parameter(leaf, torch.nn.Parameter(tensor))
            else:
                node.register_buffer(leaf, tensor)

        # edge_norm (nequip EdgeLengthNormalizer): per-type-pair cutoffs
        rmax = torch.full((T, T), float(r_max), dtype=dt)
        if per_edge_type_cutoff is not None:
            for ci, cn in enumerate(self.type_names):
                if cn not in per_edge_type_cutoff:
                    continue
                v = per_edge_type_cutoff[cn]
                for ni, nn_ in enumerate(self.type_names):
                    if isinstance(v, dict):
                        if nn_ in v:
                            rmax[ci, ni] = float(v[nn_])
                    else:
                        rmax[ci, ni] = float(v)
        add("edge_norm.rmax_recip", 1.0 / rmax, "buffer")
        if self.embed_kind == 1:
            # PerClassSpline (spline.py:43-62) + the init of TwoBodySplineScalarEmbed (scalarembed.py:136-145)
            lower = torch.arange(-span, B - span, dtype=dt) / B
            add("radial_chemical_embed.spline.lower", lower, "buffer")
            add("radial_chemical_embed.spline.upper", lower + (span + 1) / B, "buffer")
            bound = math.sqrt(3 / span) if forward_normalize else math.sqrt(3 / S0)
            add("radial_chemical_embed.spline.class_embed.weight",
                torch.empty(T * T, S0 * B, dtype=dt).uniform_(-bound, bound), "param")
        else:
            # nequip's BesselEdgeLengthEncoding (EXT) holds its roots as a Parameter when `trainable`, as a buffer otherwise
            add("radial_chemical_embed.bessel_encode.bessel_weights",
                (torch.linspace(1.0, B, B, dtype=dt) * math.pi).unsqueeze(0), "param" if rce.get("bessel_trainable", False) else "buffer")
            add("radial_chemical_embed.type_embed.center_embed.weight", torch.randn(T, S0 // 2, dtype=dt), "param")
            add("radial_chemical_embed.type_embed.neighbor_embed.weight", torch.randn(T, S0 // 2, dtype=dt), "param")
            add("radial_chemical_embed.type_embed.basis_linear.mlp.0.weight", rng_u(B, S0), "param")

        def add_mlp(prefix, dims):
            for i, (a, b) in enumerate(zip(dims, dims[1:])):
                add(f"{prefix}.{i}.weight", rng_u(a, b), "param")

        R = l_max + 1
        W = R * u
        We = W if weight_individual_irreps else u  # env-weight columns of the Allegro layers (_channels.py:29-35)
        self._embed_dims = [S0] + [scalar_embed_mlp_hidden_layers_width] * scalar_embed_mlp_hidden_layers_depth + [S]
        add_mlp("scalar_embed_mlp.mlp.mlp", self._embed_dims)
        add_mlp("tensor_embed.env_embed_linear.mlp", [S, W])
        if not weight_individual_irreps:
            # the reference registers an empty, persistent `_rtoi` in this mode (_channels.py:29-31): keep the key
            add("allegro._env_weighter._rtoi", torch.empty(0, dtype=dt), "buffer")
        add_mlp("allegro.first_layer_env_embed_projection.mlp", [S, S + We])
        self.tps_irreps = allegro_layer_irreps(l_max, parity, L)
        env = o3.Irreps.spherical_harmonics(l_max, p=-1)
        self._tp_meta = []
        for l in range(L):
            lat_dims = [S * (l + 1) + u] + [allegro_mlp_hidden_layers_width] * allegro_mlp_hidden_layers_depth + \
                       [S + (We if l < L - 1 else 0)]
            add_mlp(f"allegro.latents.{l}.mlp", lat_dims)
        for l in range(L):
            w3j, instr, diag, dims = build_w3j(self.tps_irreps[l], env, self.tps_irreps[l + 1])
            P = len(instr)
            shape = ((u,) if tp_path_channel_coupling else tuple()) + ((P,) if P > 1 else tuple())
            add(f"allegro.tps.{l}.weights", rng_u(*shape) if len(shape) else rng_u(1).reshape(()), "param")
            add(f"allegro.tps.{l}.w3j", torch.tensor(w3j, dtype=dt), "buffer")
            self._tp_meta.append(dict(num_paths=P, diag=diag, dims=dims))
        ro_dims = [S * (L + 1)] + [readout_mlp_hidden_layers_width] * readout_mlp_hidden_layers_depth + [1]
        add_mlp("edge_readout.mlp.mlp", ro_dims)

        def per_type(v):
            if v is None:
                return None
            if isinstance(v, dict):
                v = [v[t] for t in self.type_names]
            t = torch.as_tensor(v, dtype=dt).reshape(-1)
            return t.expand(T).clone() if t.numel() == 1 else t

        sc, sh = per_type(per_type_energy_scales), per_type(per_type_energy_shifts)
        self.has_scales, self.has_shifts = sc is not None, sh is not None
        if sc is not None:
            add("per_type_energy_scale_shift.scales", sc, "param")
        if sh is not None:
            add("per_type_energy_scale_shift.shifts", sh, "param")
        # eval mode (the default state of a freshly built model here): the hand-written inference / force pipeline, no weight
        # gradients (like _flashallegro.py:660).  `.train()` switches every trainable parameter on and `forward` to the
        # differentiable evaluation of allegro_amd/training.py.
        self._frozen_keys = set()
        if self.has_scales and not per_type_energy_scales_trainable:
            self._frozen_keys.add("func.per_type_energy_scale_shift.scales")
        if self.has_shifts and not per_type_energy_shifts_trainable:
            self._frozen_keys.add("func.per_type_energy_scale_shift.shifts")
        self._trainer = None
        self.training = False
        for p in self.parameters():
            p.requires_grad_(False)
        self._bound_lib: Optional[_lib.AllegroLib] = None
        # plans own device-resident tables and the packed weights / workspace are device buffers: one set per device,
        # `_plan_handle` / `_blob` / `_workspace` always refer to the device of the current call (`_select_device`)
        self._per_device: Dict[int, dict] = {}
        self._plan_handle = None
        self._plan_keep = None
        self._blob: Optional[torch.Tensor] = None
        self._blob_key = None
        self._workspace: Optional[torch.Tensor] = None
        self._cur_dev = None
        self._graph_cache = None  # (key, keyed tensors kept alive, PreparedGraph structure) of forward()

    # -- training ---------------------------------------------------------------------------------
    def train(self, mode: bool = True):
        """Training mode: every parameter the reference trains requires grad (MLPs, type embeddings, tensor-product path
        weights; per-type scales / shifts only with `per_type_energy_*_trainable`, allegro_models.py:251-260) and `forward`
        returns energies, forces and stress attached to the autograd graph (allegro_amd/training.py).  `.eval()` returns to
        the inference pipeline; the packed device weights follow in-place optimizer updates (`_ensure_weights`)."""
        mode = bool(mode)
        was = self.training
        super().train(mode)
        if mode == was:
            return self  # (a repeated .train() / .eval(), as training loops issue every epoch, leaves the user's freezes alone)
        if mode:
            # the parameters that were trainable when training mode was last left (so a user's requires_grad_(False) on a
            # sub-module survives an eval() / train() round trip); the first time: everything the reference trains
            on = getattr(self, "_trainable_keys", None)
            for k, p in self.named_parameters():
                p.requires_grad_(k not in self._frozen_keys if on is None else k in on)
        else:
            self._trainable_keys = {k for k, p in self.named_parameters() if p.requires_grad}
            for p in self.parameters():
                p.requires_grad_(False)
        return self

    def _training_evaluator(self):
        if self._trainer is None:
            from .training import TrainingEvaluator

            self._trainer = TrainingEvaluator(self)
        return self._trainer

    def chunked_training_step(self, graph: "PreparedGraph", max_edges_per_chunk: int):
        """Training on boxes whose whole differentiable graph does not fit: `step(pos, loss_fn)` of the returned object
        accumulates the exact gradient of `loss_fn(forces, total_energy)` block of center atoms by block
        (allegro_amd/training.py: ChunkedTrainingStep; peak memory ~ the block, not the frame)."""
        from .training import ChunkedTrainingStep

        return ChunkedTrainingStep(self._training_evaluator(), graph, max_edges_per_chunk)

    # -- library / plan -------------------------------------------------------------------------
    def _get_lib(self) -> _lib.AllegroLib:
        return self._bound_lib if self._bound_lib is not None else _lib.load()

    def _bind_library(self, lib: _lib.AllegroLib):
        """(tests) bind an explicitly loaded library instead of the default gfx950 one."""
        self._drop_device_state()
        self._bound_lib = lib
        self._trainer = None

    def _drop_device_state(self):
        self._stash_device_state()
        for st in self._per_device.values():
            if st.get("plan") is not None:
                try:
                    self._get_lib().model_plan_destroy(st["plan"])
                except Exception:
                    pass
        self._per_device = {}
        self._plan_handle = self._plan_keep = self._blob = self._blob_key = self._workspace = self._cur_dev = None

    def _stash_device_state(self):
        if self._cur_dev is not None:
            self._per_device[self._cur_dev] = dict(plan=self._plan_handle, keep=self._plan_keep, blob=self._blob,
                                                   blob_key=self._blob_key, ws=self._workspace,
                                                   hip_graph=getattr(self, "_hip_graph", False), out=getattr(self, "_out", None))

    def _select_device(self, device) -> None:
        """Switch the per-device state (plan, packed weights, workspace) to `device`."""
        device = torch.device(device)
        idx = device.index if device.type == "cuda" and device.index is not None else (
            torch.cuda.current_device() if device.type == "cuda" else -1)
        if idx == self._cur_dev:
            return
        self._stash_device_state()
        st = self._per_device.get(idx, {})
        self._plan_handle, self._plan_keep = st.get("plan"), st.get("keep")
        self._blob, self._blob_key, self._workspace = st.get("blob"), st.get("blob_key"), st.get("ws")
        self._hip_graph, self._out = st.get("hip_graph", False), st.get("out")
        self._cur_dev = idx

    def _sd(self) -> Dict[str, torch.Tensor]:
        return {k[len("func."):]: v for k, v in self.state_dict().items()}

    def _ensure_plan(self):
        if self._cur_dev is None:  # first use without tensors (export, enable_hip_graph): the current device
            self._select_device(torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                else torch.device("cpu"))
        if self._plan_handle is not None:
            return
        cfg, keep = self._build_config()
        self._plan_handle = self._get_lib().model_plan_create(cfg)
        self._plan_keep = (cfg, keep)

    def _build_config(self):
        """(`aa_model_config`, keep-alive list) of this model: hyper-parameters + the Clebsch-Gordan non-zeros of every layer.
        Pure host work (no library call)."""
        hp = self.hparams
        cfg = _lib.ModelConfig()
        cfg.dtype = _TORCH2AA[self.dtype]
        cfg.num_types = len(self.type_names)
        cfg.num_bessels, cfg.poly_p = hp["num_bessels"], hp["poly_p"]
        cfg.l_max, cfg.num_layers = hp["l_max"], hp["num_layers"]
        cfg.num_scalar, cfg.num_tensor, cfg.embed_dim = hp["num_scalar_features"], hp["num_tensor_features"], hp["embed_dim"]
        cfg.embed_mlp_depth, cfg.embed_mlp_width = hp["embed_depth"], hp["embed_width"]
        cfg.latent_mlp_depth, cfg.latent_mlp_width = hp["latent_depth"], hp["latent_width"]
        cfg.readout_mlp_depth, cfg.readout_mlp_width = hp["readout_depth"], hp["readout_width"]
        cfg.forward_weight_init = int(hp["forward_normalize"])
        cfg.avg_num_neighbors = hp["avg_num_neighbors"]
        cfg.act_const = silu_second_moment_const()
        cfg.env_shared_weights = int(not self.weight_individual_irreps)
        for i, nl in enumerate(self.nonlinearities):
            cfg.act_kind[i] = ACT_KINDS[nl]
            cfg.act_consts[i] = second_moment_const(nl)
        cfg.has_scales, cfg.has_shifts = int(self.has_scales), int(self.has_shifts)
        cfg.embed_kind, cfg.spline_span = self.embed_kind, hp["spline_span"]
        cfg.bessel_convention = _BESSEL_CONVENTIONS[self.bessel_convention]
        keep = []
        sd = self._sd()
        for l in range(hp["num_layers"]):
            m = self._tp_meta[l]
            desc, k = w3j_to_desc(sd[f"allegro.tps.{l}.w3j"], hp["num_tensor_features"], m["dims"], m["num_paths"],
                                  m["diag"], hp["coupling"])
            cfg.tps[l] = desc
            keep.append(k)
        return cfg, keep

    def _ensure_weights(self, device):
        key = (str(device), tuple(int(p._version) for p in self.parameters()), tuple(p.data_ptr() for p in self.parameters()))
        if self._blob is not None and self._blob_key == key:
            return
        self._blob, self._blob_key = self._pack_blob(self._plan_handle, device), key

    def _raw_tensors(self, sd=None):
        """[(slot, float64 tensor)]: the parameters aa_model_pack_weights consumes, in the reference's own state_dict layout; slot =
        position of the pointer in `aa_model_raw_weights` (include/allegro_amd.h) -- also the tensor ids of a host model file
        (allegro_amd.export.write_host_model, csrc/aa_hostfile.hip)."""
        hp = self.hparams
        if sd is None:
            sd = {k: v.detach().to("cpu", torch.float64).contiguous() for k, v in self._sd().items()}
        ML, LL = _lib.AA_MAX_MLP_LAYERS, _lib.AA_MAX_LAYERS
        T = len(self.type_names)
        rr = sd["edge_norm.rmax_recip"]
        out = [(0, rr.expand(T, T).contiguous() if rr.numel() == 1 else rr)]
        if self.embed_kind == 1:
            out.append((7 + 2 * ML + LL * ML + LL + 2, sd["radial_chemical_embed.spline.class_embed.weight"]))
        else:
            out += [(1, sd["radial_chemical_embed.bessel_encode.bessel_weights"]),
                    (2, sd["radial_chemical_embed.type_embed.center_embed.weight"]),
                    (3, sd["radial_chemical_embed.type_embed.neighbor_embed.weight"]),
                    (4, sd["radial_chemical_embed.type_embed.basis_linear.mlp.0.weight"])]
        out += [(5 + i, sd[f"scalar_embed_mlp.mlp.mlp.{i}.weight"]) for i in range(hp["embed_depth"] + 1)]
        out += [(5 + ML, sd["tensor_embed.env_embed_linear.mlp.0.weight"]),
                (6 + ML, sd["allegro.first_layer_env_embed_projection.mlp.0.weight"])]
        for l in range(hp["num_layers"]):
            out += [(7 + ML + l * ML + i, sd[f"allegro.latents.{l}.mlp.{i}.weight"]) for i in range(hp["latent_depth"] + 1)]
            out.append((7 + ML + LL * ML + l, sd[f"allegro.tps.{l}.weights"]))
        out += [(7 + ML + LL * ML + LL + i, sd[f"edge_readout.mlp.mlp.{i}.weight"]) for i in range(hp["readout_depth"] + 1)]
        if self.has_scales:
            out.append((7 + 2 * ML + LL * ML + LL, sd["per_type_energy_scale_shift.scales"]))
        if self.has_shifts:
            out.append((7 + 2 * ML + LL * ML + LL + 1, sd["per_type_energy_scale_shift.shifts"]))
        return sorted(out, key=lambda kv: kv[0])

    def _pack_blob(self, plan_handle, device) -> torch.Tensor:
        """The model's state_dict packed for `plan_handle` (aa_model_pack_weights): the blob's layout belongs to that plan
        (aa_model_plan_layout_hash)."""
        lib = self._get_lib()
        hp = self.hparams
        sd = {k: v.detach().to("cpu", torch.float64).contiguous() for k, v in self._sd().items()}
        keep = []

        def ptr(t):
            a = np.ascontiguousarray(t.numpy().reshape(-1))
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))

        raw = _lib.RawWeights()
        ML, LL = _lib.AA_MAX_MLP_LAYERS, _lib.AA_MAX_LAYERS
        for slot, t in self._raw_tensors(sd):
            p = ptr(t)
            if slot < 5:
                setattr(raw, ("rmax_recip", "bessel_weights", "center_embed", "neighbor_embed", "basis_linear")[slot], p)
            elif slot < 5 + ML:
                raw.embed_mlp[slot - 5] = p
            elif slot < 7 + ML:
                setattr(raw, ("env_embed_linear", "first_proj")[slot - 5 - ML], p)
            elif slot < 7 + ML + LL * ML:
                raw.latent[(slot - 7 - ML) // ML][(slot - 7 - ML) % ML] = p
            elif slot < 7 + ML + LL * ML + LL:
                raw.tp_weights[slot - 7 - ML - LL * ML] = p
            elif slot < 7 + 2 * ML + LL * ML + LL:
                raw.readout[slot - 7 - ML - LL * ML - LL] = p
            else:
                setattr(raw, ("scales", "shifts", "spline_weights")[slot - 7 - 2 * ML - LL * ML - LL], p)
        nbytes = lib.lib.aa_model_weights_bytes(plan_handle)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
        lib.check(lib.lib.aa_model_pack_weights(plan_handle, C.byref(raw), blob.data_ptr(), nbytes, stream),
                  "aa_model_pack_weights")
        return blob

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        # w3j buffers may have changed: rebuild device tables lazily
        self._drop_device_state()
        self._trainer = None
        return out

    # -- evaluation -----------------------------------------------------------------------------
    def prepare_graph(self, edge_index, atom_types, num_atoms, shift_vec=None) -> PreparedGraph:
        return PreparedGraph(edge_index, atom_types, num_atoms, shift_vec, lib=self._bound_lib)

    def energy_forces(self, pos: torch.Tensor, graph: PreparedGraph, with_forces: bool = True):
        """One pass of the hot path: returns (atom_energy [N], forces [N,3] | None)."""
        lib = self._get_lib()
        _require_gpu(lib, pos, "HipAllegroModel")
        assert pos.dtype == self.dtype, f"positions must be {self.dtype}"
        with _device_ctx(pos.device):
            return self._energy_forces_on_device(lib, pos, graph, with_forces)

    def _energy_forces_on_device(self, lib, pos, graph, with_forces):
        self._select_device(pos.device)
        self._ensure_plan()
        self._ensure_weights(pos.device)
        N, E = graph.num_atoms, graph.num_edges
        need = lib.lib.aa_model_workspace_bytes(self._plan_handle, N, E, int(with_forces))
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != pos.device:
            # The arena is the caller's (C ABI): this host grows it with 6 % of headroom and drops the old one FIRST.  In an MD loop
            # the edge count creeps up and down with every neighbour list; an arena sized exactly for the largest list so far is
            # re-allocated at every new maximum, and a fresh 30-GB device allocation costs ~0.5 s on this stack (two such steps in
            # the first 25 fs of the C4 run of tools/md_loop.py: profiles/r05_v22_md_loop_c4.json) while old + new briefly coexist.
            self._workspace = None
            self._workspace = torch.empty(need + need // 16 + (1 << 20), dtype=torch.uint8, device=pos.device)
        pos = pos.detach().contiguous()
        if getattr(self, "_hip_graph", False):
            # replay mode: outputs live in persistent buffers so that every argument of the step keeps its address
            # (the caller updates `pos` in place and reads the results before the next call)
            if getattr(self, "_out", None) is None or self._out[0].shape[0] != N or self._out[0].device != pos.device:
                self._out = (torch.empty(N, dtype=self.dtype, device=pos.device),
                             torch.empty((N, 3), dtype=self.dtype, device=pos.device))
            e_atom, forces = self._out[0], (self._out[1] if with_forces else None)
        else:
            e_atom = torch.empty(N, dtype=self.dtype, device=pos.device)
            forces = torch.empty((N, 3), dtype=self.dtype, device=pos.device) if with_forces else None
        g = graph.c_struct()
        lib.check(lib.lib.aa_model_energy_forces(self._plan_handle, self._blob.data_ptr(), C.byref(g), pos.data_ptr(),
                                                 self._workspace.data_ptr(), self._workspace.numel(), e_atom.data_ptr(),
                                                 forces.data_ptr() if with_forces else None, _stream_ptr(pos)),
                  "aa_model_energy_forces")
        return e_atom, forces

    def describe_plan(self) -> dict:
        """`aa_model_plan_describe`: which forward / folds the plan of the current device runs (benchmark lines, bug reports)."""
        import json

        lib = self._get_lib()
        self._ensure_plan()
        buf = C.create_string_buffer(1024)
        lib.check(min(0, lib.lib.aa_model_plan_describe(self._plan_handle, buf, 1024)), "aa_model_plan_describe")
        return json.loads(buf.value.decode())

    def check(self, device=None) -> None:
        """`aa_model_check`: waits for the steps enqueued so far and raises if one of them contradicted the hints of its graph
        (a `max_degree` smaller than a real segment, edges outside the atom block).  `PreparedGraph` derives both hints from
        the row pointers, so this can only fire for hand-made graphs -- the raw C ABI has the same guard."""
        lib = self._get_lib()
        if getattr(self, "_plan_handle", None) is None:
            return
        dev = device if device is not None else (self._workspace.device if self._workspace is not None else None)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev is not None and dev.type == "cuda" else None
        lib.check(lib.lib.aa_model_check(self._plan_handle, stream), "aa_model_check")

    def virial(self, graph: PreparedGraph) -> torch.Tensor:
        """dE/d(strain) [3,3] of the LAST `energy_forces(..., with_forces=True)` call on `graph` (stress = virial / volume,
        nequip ForceStressOutput; LAMMPS' virial is its negative)."""
        lib = self._get_lib()
        out = torch.empty(9, dtype=self.dtype, device=self._workspace.device)
        g = graph.c_struct()
        with _device_ctx(out.device):
            return self._virial_on_device(lib, g, out)

    def _virial_on_device(self, lib, g, out):
        lib.check(lib.lib.aa_model_virial(self._plan_handle, C.byref(g), self._workspace.data_ptr(), self._workspace.numel(),
                                          out.data_ptr(), _stream_ptr(out)), "aa_model_virial")
        return out.view(3, 3)

    def enable_hip_graph(self, on: bool = True) -> None:
        """Capture the step's launch sequence into a hipGraph and replay it (aa_model_plan_enable_graph): for
        launch-bound small systems in MD loops.  `pos` must then be updated in place between calls."""
        lib = self._get_lib()
        self._ensure_plan()
        lib.check(lib.lib.aa_model_plan_enable_graph(self._plan_handle, int(on)), "aa_model_plan_enable_graph")
        self._hip_graph = bool(on)
        self._out = None

    def enable_debug_taps(self, on: bool = True) -> None:
        """Make the following steps materialise the per-edge intermediates `debug_tap` reads (staged pipeline instead
        of the fused kernels, which keep them on chip).  Parity tests only."""
        lib = self._get_lib()
        self._ensure_plan()
        lib.check(lib.lib.aa_model_plan_enable_taps(self._plan_handle, int(on)), "aa_model_plan_enable_taps")

    def debug_tap(self, name: str, graph: PreparedGraph, with_forces: bool = False) -> torch.Tensor:
        """Copy of a per-edge intermediate of the LAST step out of the workspace ("dvec" / "vec" need the layout of a
        step that computed forces: `with_forces=True`)."""
        lib = self._get_lib()
        ptr, ld = C.c_void_p(), C.c_int64()
        rc = lib.lib.aa_model_debug_tap(self._plan_handle, (name + ("+f" if with_forces else "")).encode(), graph.num_atoms,
                                        graph.num_edges, self._workspace.data_ptr(), C.byref(ptr), C.byref(ld))
        if rc < 0:
            lib.check(rc, "aa_model_debug_tap")
        off = (ptr.value - self._workspace.data_ptr()) // self._workspace.element_size()
        esz = 4 if self.dtype == torch.float32 else 8
        flat = self._workspace[off: off + graph.num_edges * ld.value * esz].view(self.dtype)
        return flat.view(graph.num_edges, ld.value).clone()

    def _graph_for(self, data: Dict[str, torch.Tensor]) -> PreparedGraph:
        """Center-sorted CSR view of `data`'s edge list.  The STRUCTURE (sort permutation, rowptr, transposed CSR,
        types) is cached, keyed on the identity + in-place version of `edge_index` and `atom_types`; the keyed
        tensors are kept alive by the cache entry so their addresses cannot be recycled.  Everything that depends
        on values that change while the edge list stays put -- the periodic shift vectors `edge_cell_shift @ cell`
        -- is recomputed on every call (NPT / strain scans change `cell` with a fixed edge list)."""
        pos, ei, at = data["pos"], data["edge_index"], data["atom_types"]
        key = (ei.data_ptr(), tuple(ei.shape), int(ei._version), at.data_ptr(), int(at._version), int(pos.shape[0]),
               str(ei.device))
        if self._graph_cache is None or self._graph_cache[0] != key:
            self._graph_cache = (key, (ei, at), PreparedGraph(ei, at, pos.shape[0], None))
        graph = self._graph_cache[2]
        shift_vec = None
        if "edge_cell_shift" in data and "cell" in data:
            ecs = data["edge_cell_shift"].to(self.dtype)
            cell = data["cell"].to(self.dtype).reshape(-1, 3, 3)
            if cell.shape[0] == 1:
                shift_vec = ecs @ cell[0]
            else:  # batched frames: the cell of each edge's frame (nequip with_edge_vectors_: cell[batch[center]])
                if "batch" not in data:
                    raise ValueError("a [B,3,3] cell needs the `batch` vector")
                shift_vec = torch.einsum("ei,eij->ej", ecs, cell[data["batch"][ei[0]]])
            if graph.perm is not None:
                shift_vec = shift_vec[graph.perm]
            shift_vec = shift_vec.contiguous()
        graph.shift_vec = shift_vec
        return graph

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """AtomicDataDict in, AtomicDataDict out (keys: pos, edge_index, atom_types [, cell, edge_cell_shift, batch])."""
        pos = data["pos"]
        graph = self._graph_for(data)
        if self.training and torch.is_grad_enabled():
            return self._training_evaluator().forward(data, graph)
        e_atom, forces = self.energy_forces(pos.to(self.dtype), graph, with_forces=True)
        out = dict(data)
        out["atomic_energy"] = e_atom.unsqueeze(-1)
        nf = 1
        if "batch" in data:
            nf = int(data["cell"].reshape(-1, 3, 3).shape[0]) if "cell" in data else int(data["batch"].max()) + 1
            out["total_energy"] = torch.zeros(nf, 1, dtype=self.dtype, device=pos.device).index_add_(0, data["batch"], e_atom.unsqueeze(-1))
        else:
            out["total_energy"] = e_atom.sum().reshape(1, 1)
        out["forces"] = forces
        if "cell" in data:
            # nequip ForceStressOutput (EXT) conventions: stress = dE/d(strain) / volume, virial = -dE/d(strain)
            cell = data["cell"].to(self.dtype).reshape(-1, 3, 3)
            if nf == 1:
                w = self.virial(graph).unsqueeze(0)
            else:
                # per-frame strain derivative from the per-edge dE/dr_e and r_e the step left in the workspace
                d = self.debug_tap("dvec", graph, with_forces=True)[:, :3]
                v = self.debug_tap("vec", graph, with_forces=True)
                r = v[:, :3] * v[:, 3:4]
                frame = data["batch"][graph.center.long()]
                w = torch.zeros(nf, 3, 3, dtype=self.dtype, device=pos.device).index_add_(
                    0, frame, d.unsqueeze(2) * r.unsqueeze(1))
            vol = torch.linalg.det(cell).abs().reshape(-1, 1, 1)
            out["stress"] = w / vol
            out["virial"] = -w
        return out

    def __del__(self):
        try:
            self._drop_device_state()
        except Exception:
            pass
