"""`torch.library` registration of the tensor-product operator (SURVEY §8b "operator registration").

The reference registers its accelerated contraction as a functional library op with a fake (meta) kernel and
`register_autograd` (allegro/nn/_strided/_flashallegro.py:489,533-670) so it composes with `torch.compile` /
`torch.export`.  The same contract here, around the C ABI (`aa_tp_forward` / `aa_tp_backward`):

    torch.ops.allegro_amd.tp_forward (x1, x2, weights, rowptr, eids?, num_atoms, scatter_factor, plan, lib_id, d2, dout)
        -> (out [E,u,dout], x2s [N,u,d2])
    torch.ops.allegro_amd.tp_backward(gout, x1, x2s, weights, rowptr, eids?, num_atoms, scatter_factor, plan, lib_id)
        -> (gx1 [E,u,d1], gx2 [E,u,d2])

    torch.ops.allegro_amd.tp_backward_weights(gout, x1, x2s, weights, rowptr, eids?, num_atoms, plan, lib_id)
        -> gweights (shape of weights)
    torch.ops.allegro_amd.tp_backward_x1(gout, x2s, weights, ...) -> gx1      (one gradient alone: `aa_tp_backward` with
    torch.ops.allegro_amd.tp_backward_x2(gout, x1,  weights, ...) -> gx2       the other output NULL; the training path)
    torch.ops.allegro_amd.segment_sum(x [E,u,d], rowptr, eids?, num_atoms, scale, lib_id) -> [N,u,d]

All are functional (fresh outputs, nothing mutated); all non-tensor arguments are int/float.  Unlike the reference's
Triton op, which returns `None` for the weights (`_flashallegro.py:641-666`) and therefore has to fall back to the
eager contraction in training mode (`:725-755`), the backward here also returns the path-weight gradient whenever the
weights require grad -- the coverage of the reference's eager / cuEquivariance contracters (`_contract.py:172-177`).  `plan` is the
`aa_tp_plan*` handle as an integer, `lib_id` selects the loaded library (0 = the gfx950 build; tests register the
emulation build under another id).  There is no CPU implementation: the ops raise on CPU tensors.
"""
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

_LIBS: Dict[int, "_lib.AllegroLib"] = {}


def register_library(lib: "_lib.AllegroLib") -> int:
    """Make `lib` addressable from the ops; returns its id (idempotent)."""
    for k, v in _LIBS.items():
        if v is lib:
            return k
    k = (max(_LIBS) + 1) if _LIBS else 1
    _LIBS[k] = lib
    return k


def _resolve(lib_id: int) -> "_lib.AllegroLib":
    return _lib.load() if lib_id == 0 else _LIBS[lib_id]


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _check_device(lib, t: torch.Tensor, what: str):
    if not t.is_cuda and not lib.is_emulation:
        raise _lib.AllegroError(f"{what}: tensors must live on the GPU (got {t.device}); there is no CPU fallback")


@torch.library.custom_op("allegro_amd::tp_forward", mutates_args=())
def tp_forward(x1: torch.Tensor, x2: torch.Tensor, weights: torch.Tensor, rowptr: torch.Tensor,
               eids: Optional[torch.Tensor], num_atoms: int, scatter_factor: float, plan: int, lib_id: int,
               d2: int, dout: int) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _resolve(lib_id)
    _check_device(lib, x1, "allegro_amd::tp_forward")
    x1c, x2c, wc = x1.contiguous(), x2.contiguous(), weights.contiguous()
    E, u = x1c.shape[0], x1c.shape[1]
    out = torch.empty((E, u, dout), dtype=x1.dtype, device=x1.device)
    x2s = torch.empty((num_atoms, u, d2), dtype=x1.dtype, device=x1.device)
    lib.tp_forward(plan, E, num_atoms, x1c.data_ptr(), x2c.data_ptr(), wc.data_ptr(), rowptr.data_ptr(),
                   eids.data_ptr() if eids is not None else None, scatter_factor, x2s.data_ptr(), out.data_ptr(),
                   _stream_ptr(x1))
    return out, x2s


@tp_forward.register_fake
def _(x1, x2, weights, rowptr, eids, num_atoms, scatter_factor, plan, lib_id, d2, dout):
    return x1.new_empty((x1.shape[0], x1.shape[1], dout)), x1.new_empty((num_atoms, x1.shape[1], d2))


@torch.library.custom_op("allegro_amd::tp_backward", mutates_args=())
def tp_backward(gout: torch.Tensor, x1: torch.Tensor, x2s: torch.Tensor, weights: torch.Tensor, rowptr: torch.Tensor,
                eids: Optional[torch.Tensor], num_atoms: int, scatter_factor: float, plan: int,
                lib_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _resolve(lib_id)
    _check_device(lib, x1, "allegro_amd::tp_backward")
    goutc, x1c, x2sc, wc = gout.contiguous(), x1.contiguous(), x2s.contiguous(), weights.contiguous()
    E, u = x1c.shape[0], x1c.shape[1]
    gx1 = torch.empty_like(x1c)
    gx2 = torch.empty((E, u, x2sc.shape[2]), dtype=x1.dtype, device=x1.device)
    lib.tp_backward(plan, E, num_atoms, x1c.data_ptr(), x2sc.data_ptr(), wc.data_ptr(), rowptr.data_ptr(),
                    eids.data_ptr() if eids is not None else None, scatter_factor, goutc.data_ptr(), gx1.data_ptr(),
                    gx2.data_ptr(), _stream_ptr(x1))
    return gx1, gx2


@tp_backward.register_fake
def _(gout, x1, x2s, weights, rowptr, eids, num_atoms, scatter_factor, plan, lib_id):
    return torch.empty_like(x1), x1.new_empty((x1.shape[0], x1.shape[1], x2s.shape[2]))


@torch.library.custom_op("allegro_amd::tp_backward_x1", mutates_args=())
def tp_backward_x1(gout: torch.Tensor, x2s: torch.Tensor, weights: torch.Tensor, rowptr: torch.Tensor, eids: Optional[torch.Tensor],
                   num_atoms: int, scatter_factor: float, plan: int, lib_id: int, d1: int) -> torch.Tensor:
    lib = _resolve(lib_id)
    _check_device(lib, gout, "allegro_amd::tp_backward_x1")
    goutc, x2sc, wc = gout.contiguous(), x2s.contiguous(), weights.contiguous()
    E, u = goutc.shape[0], goutc.shape[1]
    gx1 = torch.empty((E, u, d1), dtype=gout.dtype, device=gout.device)
    lib.tp_backward(plan, E, num_atoms, None, x2sc.data_ptr(), wc.data_ptr(), rowptr.data_ptr(),
                    eids.data_ptr() if eids is not None else None, scatter_factor, goutc.data_ptr(), gx1.data_ptr(), None,
                    _stream_ptr(gout))
    return gx1


@tp_backward_x1.register_fake
def _(gout, x2s, weights, rowptr, eids, num_atoms, scatter_factor, plan, lib_id, d1):
    return gout.new_empty((gout.shape[0], gout.shape[1], d1))


@torch.library.custom_op("allegro_amd::tp_backward_x2", mutates_args=())
def tp_backward_x2(gout: torch.Tensor, x1: torch.Tensor, weights: torch.Tensor, rowptr: torch.Tensor, eids: Optional[torch.Tensor],
                   num_atoms: int, scatter_factor: float, plan: int, lib_id: int, d2: int) -> torch.Tensor:
    lib = _resolve(lib_id)
    _check_device(lib, gout, "allegro_amd::tp_backward_x2")
    goutc, x1c, wc = gout.contiguous(), x1.contiguous(), weights.contiguous()
    E, u = goutc.shape[0], goutc.shape[1]
    gx2 = torch.empty((E, u, d2), dtype=gout.dtype, device=gout.device)
    lib.tp_backward(plan, E, num_atoms, x1c.data_ptr(), None, wc.data_ptr(), rowptr.data_ptr(),
                    eids.data_ptr() if eids is not None else None, scatter_factor, goutc.data_ptr(), None, gx2.data_ptr(),
                    _stream_ptr(gout))
    return gx2


@tp_backward_x2.register_fake
def _(gout, x1, weights, rowptr, eids, num_atoms, scatter_factor, plan, lib_id, d2):
    return gout.new_empty((gout.shape[0], gout.shape[1], d2))


@torch.library.custom_op("allegro_amd::segment_sum", mutates_args=())
def segment_sum(x: torch.Tensor, rowptr: torch.Tensor, eids: Optional[torch.Tensor], num_atoms: int, scale: float,
                lib_id: int) -> torch.Tensor:
    """scale + scatter-sum of `Contracter.forward` alone (allegro/nn/_strided/_contract.py:195-204), deterministic."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::segment_sum")
    if x.dtype not in (torch.float32, torch.float64):
        raise _lib.AllegroError(f"allegro_amd::segment_sum: dtype {x.dtype}")
    xc = x.contiguous()
    E = xc.shape[0]
    out = torch.empty((num_atoms,) + tuple(xc.shape[1:]), dtype=x.dtype, device=x.device)
    row = xc[0].numel() if E else int(out[0].numel()) if num_atoms else 0
    lib.tp_segment_sum(_lib.AA_F32 if x.dtype == torch.float32 else _lib.AA_F64, E, num_atoms, row, xc.data_ptr() if E else None,
                       rowptr.data_ptr(), eids.data_ptr() if eids is not None else None, scale, out.data_ptr() if num_atoms else None,
                       _stream_ptr(x))
    return out


@segment_sum.register_fake
def _(x, rowptr, eids, num_atoms, scale, lib_id):
    return x.new_empty((num_atoms,) + tuple(x.shape[1:]))


@torch.library.custom_op("allegro_amd::tp_backward_weights", mutates_args=())
def tp_backward_weights(gout: torch.Tensor, x1: torch.Tensor, x2s: torch.Tensor, weights: torch.Tensor,
                        rowptr: torch.Tensor, eids: Optional[torch.Tensor], num_atoms: int, plan: int,
                        lib_id: int) -> torch.Tensor:
    lib = _resolve(lib_id)
    _check_device(lib, x1, "allegro_amd::tp_backward_weights")
    goutc, x1c, x2sc = gout.contiguous(), x1.contiguous(), x2s.contiguous()
    E = x1c.shape[0]
    gw = torch.empty(weights.shape, dtype=x1.dtype, device=x1.device)
    nbytes = lib.lib.aa_tp_weights_workspace_bytes(plan, num_atoms)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x1.device)
    lib.tp_backward_weights(plan, E, num_atoms, x1c.data_ptr(), x2sc.data_ptr(), rowptr.data_ptr(),
                            eids.data_ptr() if eids is not None else None, goutc.data_ptr(), ws.data_ptr(), nbytes,
                            gw.data_ptr(), _stream_ptr(x1))
    return gw


@tp_backward_weights.register_fake
def _(gout, x1, x2s, weights, rowptr, eids, num_atoms, plan, lib_id):
    return x1.new_empty(weights.shape)


@torch.library.custom_op("allegro_amd::segments", mutates_args=())
def segments(idxs: torch.Tensor, num_segments: int, assume_sorted: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """Segment bookkeeping of `Contracter.forward`'s scatter index (allegro/nn/_strided/_contract.py:195-205) as an OPAQUE
    op, so that `torch.export` / `torch.compile(fullgraph=True)` can trace through HipContracter.forward: the bincount /
    cumsum / sort below are data dependent, but their results have static shapes -- rowptr int32 [num_segments + 1] and
    the stable sort permutation eids int32 [E] (the identity when `assume_sorted`, which the kernels accept)."""
    flat = idxs.reshape(-1)
    rowptr = torch.zeros(num_segments + 1, dtype=torch.int32, device=flat.device)
    rowptr[1:] = torch.cumsum(torch.bincount(flat, minlength=num_segments), 0).to(torch.int32)
    if assume_sorted:
        eids = torch.arange(flat.numel(), dtype=torch.int32, device=flat.device)
    else:
        eids = torch.argsort(flat, stable=True).to(torch.int32)
    return rowptr, eids


@segments.register_fake
def _(idxs, num_segments, assume_sorted):
    return (idxs.new_empty((num_segments + 1,), dtype=torch.int32), idxs.new_empty((idxs.numel(),), dtype=torch.int32))


def _setup_context(ctx, inputs, output):
    x1, _x2, weights, rowptr, eids, num_atoms, scatter_factor, plan, lib_id, _d2, _dout = inputs
    _out, x2s = output
    ctx.save_for_backward(x1, x2s, weights, rowptr, eids)
    ctx.num_atoms, ctx.scatter_factor, ctx.plan, ctx.lib_id = num_atoms, scatter_factor, plan, lib_id


def _backward(ctx, gout, _gx2s):
    x1, x2s, weights, rowptr, eids = ctx.saved_tensors
    gx1, gx2 = torch.ops.allegro_amd.tp_backward(gout, x1, x2s, weights.detach(), rowptr, eids, ctx.num_atoms,
                                                 ctx.scatter_factor, ctx.plan, ctx.lib_id)
    gw = None
    if ctx.needs_input_grad[2]:  # training: path-weight gradient (first order; the x1/x2 gradients above are what
        # forces need, and a force-matching loss differentiates THEM again -- not supported through this op)
        gw = torch.ops.allegro_amd.tp_backward_weights(gout, x1, x2s, weights.detach(), rowptr, eids, ctx.num_atoms,
                                                       ctx.plan, ctx.lib_id)
    return gx1, gx2, gw, None, None, None, None, None, None, None, None


tp_forward.register_autograd(_backward, setup_context=_setup_context)


# ------------------------------------------------------------------------------------------------------------------
# Training path: arbitrarily differentiable contraction
# ------------------------------------------------------------------------------------------------------------------
# The contraction is one trilinear form  T(a, b, c; w) = sum_p w[ch,p] sum_nz C_nz a_i b_j c_k  per (edge, channel)
# (a = x1, b = x2 gathered per edge, c = the cotangent of the output).  The forward, both input gradients and the
# path-weight gradient are its four partial contractions, and the derivative of any of them with respect to any
# operand is again one of the four with operands substituted -- so a family of four mutually recursive
# autograd Functions, each one launch of a kernel that already exists (`aa_tp_forward` / `aa_tp_backward` /
# `aa_tp_backward_weights` with every edge as its own segment), is differentiable to any order.  That is what a
# force-matching loss needs (forces are first derivatives; the loss differentiates them again with respect to the
# weights), i.e. what the reference gets from autograd through its eager Contracter in training mode
# (allegro/nn/_strided/_contract.py:213-251; its Triton op has no second-order formula and falls back to eager,
# _flashallegro.py:725-755).  The scale + scatter + gather around the contraction (_contract.py:195-205) are plain
# differentiable torch ops in this path.
class _TriCtx:
    """Static description shared by the four functions: plan handle, library id, dims -- and, for the SEGMENTED form, the
    scatter index in CSR form.  Segmented: the b operand enters through the symmetric linear map M = gather . scale .
    scatter-sum over equal scatter index (allegro/nn/_strided/_contract.py:195-205), i.e. the form is
    F(a, b, c; w) = T(a, M b, c; w).  Because M is linear and symmetric, F is again trilinear and the closure of its four
    partial contractions under differentiation is the same as T's with  K(a, b) = T_k(a, M b),  I(c, b) = T_i(c, M b),
    J(c, a) = M T_j(c, a),  W(c, a, b) = T_w(c, a, M b)  -- which are exactly what `aa_tp_forward` / `aa_tp_backward` /
    `aa_tp_backward_weights` compute on true center segments (scale + scatter + gather fused into the kernels, one
    workgroup per center atom instead of one per edge)."""

    def __init__(self, plan: int, lib_id: int, d1: int, d2: int, dout: int, segments=None):
        self.plan, self.lib_id, self.d1, self.d2, self.dout = plan, lib_id, d1, d2, dout
        self.segments = segments  # None | (rowptr int32 [N+1], eids int32 [E] | None, idxs int64 [E], num_atoms, scatter_factor)
        # M b of the b operands this contraction has seen: the forward kernel returns it, and the x1- and weight-gradient kernels of
        # every later derivative take it as an input -- recomputing it was 14 segment sums of [E,u,d2] per training step at C3 (2.3 ms)
        # for 4 distinct operands.  An entry holds a WEAK reference to its `b`: while the tensor lives its address cannot be re-used, so
        # key equality means the same memory at the same version; once autograd releases the operand ([E,u,d2]: the large one) the
        # entry is dead and the buffer is free -- a strong reference kept it until every ctx of the recorded graph died.
        self.x2s_cache = {}
        self.keep = False  # remember operands even when no graph is recorded (set for the duration of one backward call)

    @staticmethod
    def _key(b):
        return (b.data_ptr(), b._version, tuple(b.shape), tuple(b.stride()), b.dtype)

    def remember(self, b, x2s):
        import weakref

        for k in [k for k, (r, _) in self.x2s_cache.items() if r() is None]:
            del self.x2s_cache[k]
        self.x2s_cache[self._key(b)] = (weakref.ref(b), x2s)

    def lookup(self, b):
        hit = self.x2s_cache.get(self._key(b))
        if hit is None:
            return None
        if hit[0]() is None:  # (the operand is gone: whatever sits at that address now is another tensor)
            del self.x2s_cache[self._key(b)]
            return None
        return hit[1]

    def forget(self, b):
        self.x2s_cache.pop(self._key(b), None)


class _recording:
    """Inside a member's forward: remember M b of its operands iff this node is part of a recorded graph (its inputs are saved and
    stay alive until the graph is freed, so the cache extends no tensor's life).  (`torch.is_grad_enabled()` is always False there.)"""

    def __init__(self, ctx, t):
        self.t, self.on = t, any(ctx.needs_input_grad)

    def __enter__(self):
        self.prev = self.t.keep
        self.t.keep = self.prev or self.on

    def __exit__(self, *a):
        self.t.keep = self.prev


def _NO_X2S_CACHE() -> bool:  # AA_TP_NO_X2S_CACHE=1: recompute M b every time (A/B and the test of the cache)
    import os

    return os.environ.get("AA_TP_NO_X2S_CACHE", "0")[:1] == "1"


def _edge_rowptr(E: int, device) -> torch.Tensor:
    return torch.arange(E + 1, dtype=torch.int32, device=device)


def _segment_sum(t: _TriCtx, b):  # x2s = scale * scatter-sum of b over the scatter index  [N,u,d2]
    hit = None if _NO_X2S_CACHE() else t.lookup(b)
    if hit is not None:
        return hit
    rowptr, eids, _idxs, n, sf = t.segments
    x2s = torch.ops.allegro_amd.segment_sum(b.detach(), rowptr, eids, n, sf, t.lib_id)
    if t.keep:  # (only while a recorded graph keeps the operand alive anyway: see _recording)
        t.remember(b, x2s)
    return x2s


def _raw_out(t: _TriCtx, a, b, w):  # [E,u,dout]
    E = a.shape[0]
    if t.segments is not None:
        rowptr, eids, _idxs, n, sf = t.segments
        out, x2s = torch.ops.allegro_amd.tp_forward(a.detach(), b.detach(), w.detach(), rowptr, eids, n, sf, t.plan, t.lib_id, t.d2, t.dout)
        if t.keep:
            t.remember(b, x2s)
        return out
    out, _ = torch.ops.allegro_amd.tp_forward(a.detach(), b.detach(), w.detach(), _edge_rowptr(E, a.device), None, E, 1.0,
                                              t.plan, t.lib_id, t.d2, t.dout)
    return out


def _raw_grad_a(t: _TriCtx, c, b, w):  # d/d a [E,u,d1]: T with (c, b) filled -- one launch, only this gradient
    E = c.shape[0]
    if t.segments is not None:
        rowptr, eids, _idxs, n, sf = t.segments
        return torch.ops.allegro_amd.tp_backward_x1(c.detach(), _segment_sum(t, b), w.detach(), rowptr, eids, n, sf, t.plan, t.lib_id, t.d1)
    return torch.ops.allegro_amd.tp_backward_x1(c.detach(), b.detach(), w.detach(), _edge_rowptr(E, c.device), None, E, 1.0,
                                                t.plan, t.lib_id, t.d1)


def _raw_grad_b(t: _TriCtx, c, a, w):  # d/d b [E,u,d2]: T with (c, a) filled (segmented: M applied)
    E = c.shape[0]
    if t.segments is not None:
        rowptr, eids, _idxs, n, sf = t.segments
        return torch.ops.allegro_amd.tp_backward_x2(c.detach(), a.detach(), w.detach(), rowptr, eids, n, sf, t.plan, t.lib_id, t.d2)
    return torch.ops.allegro_amd.tp_backward_x2(c.detach(), a.detach(), w.detach(), _edge_rowptr(E, c.device), None, E, 1.0,
                                                t.plan, t.lib_id, t.d2)


def _raw_wgrad(t: _TriCtx, c, a, b, w_like):  # [shape of w]
    E = a.shape[0]
    if t.segments is not None:
        rowptr, eids, _idxs, n, _sf = t.segments
        return torch.ops.allegro_amd.tp_backward_weights(c.detach(), a.detach(), _segment_sum(t, b), w_like.detach(), rowptr, eids, n, t.plan, t.lib_id)
    return torch.ops.allegro_amd.tp_backward_weights(c.detach(), a.detach(), b.detach(), w_like.detach(),
                                                     _edge_rowptr(E, a.device), None, E, t.plan, t.lib_id)


class _TriK(torch.autograd.Function):
    """out[k] = T with (a, b) filled."""

    @staticmethod
    def forward(ctx, a, b, w, t):
        ctx.t = t
        ctx.save_for_backward(a, b, w)
        with _recording(ctx, t):
            return _raw_out(t, a, b, w)

    @staticmethod
    def backward(ctx, g):
        a, b, w = ctx.saved_tensors
        t = ctx.t
        n = ctx.needs_input_grad  # (only the partial contractions some gradient actually needs are launched)
        return (_TriI.apply(g, b, w, t) if n[0] else None, _TriJ.apply(g, a, w, t) if n[1] else None,
                _TriW.apply(g, a, b, w, t) if n[2] else None, None)


class _TriI(torch.autograd.Function):
    """[i]: T with (c, b) filled (the x1 gradient)."""

    @staticmethod
    def forward(ctx, c, b, w, t):
        ctx.t = t
        ctx.save_for_backward(c, b, w)
        with _recording(ctx, t):
            return _raw_grad_a(t, c, b, w)

    @staticmethod
    def backward(ctx, h):
        c, b, w = ctx.saved_tensors
        t = ctx.t
        n = ctx.needs_input_grad
        return (_TriK.apply(h, b, w, t) if n[0] else None, _TriJ.apply(c, h, w, t) if n[1] else None,
                _TriW.apply(c, h, b, w, t) if n[2] else None, None)


class _TriJ(torch.autograd.Function):
    """[j]: T with (c, a) filled (the gradient of the gathered x2)."""

    @staticmethod
    def forward(ctx, c, a, w, t):
        ctx.t = t
        ctx.save_for_backward(c, a, w)
        with _recording(ctx, t):
            return _raw_grad_b(t, c, a, w)

    @staticmethod
    def backward(ctx, h):
        c, a, w = ctx.saved_tensors
        t = ctx.t
        n = ctx.needs_input_grad
        # (h enters the b slot of up to three members: its M h once, for the duration of this call when no graph is recorded)
        temp = t.segments is not None and (n[1] or n[2]) and not torch.is_grad_enabled() and t.lookup(h) is None
        t.keep = temp  # (the forward kernel of the first member returns M h: the other two take it from the cache)
        try:
            return (_TriK.apply(a, h, w, t) if n[0] else None, _TriI.apply(c, h, w, t) if n[1] else None,
                    _TriW.apply(c, a, h, w, t) if n[2] else None, None)
        finally:
            if temp:
                t.keep = False
                t.forget(h)


class _TriW(torch.autograd.Function):
    """[u,P] / [P]: T with (c, a, b) filled, summed over edges (and channels when uncoupled)."""

    @staticmethod
    def forward(ctx, c, a, b, w_like, t):
        ctx.t = t
        ctx.save_for_backward(c, a, b)
        with _recording(ctx, t):
            return _raw_wgrad(t, c, a, b, w_like)

    @staticmethod
    def backward(ctx, hw):
        c, a, b = ctx.saved_tensors
        t = ctx.t
        n = ctx.needs_input_grad
        return (_TriK.apply(a, b, hw, t) if n[0] else None, _TriI.apply(c, b, hw, t) if n[1] else None,
                _TriJ.apply(c, a, hw, t) if n[2] else None, None, None)


def contract_differentiable(x1: torch.Tensor, x2_gathered: torch.Tensor, weights: torch.Tensor, plan: int, lib_id: int,
                            d1: int, d2: int, dout: int) -> torch.Tensor:
    """`Contracter._contract(x1, x2)` (allegro/nn/_strided/_contract.py:213-251) with derivatives of every order."""
    return _TriK.apply(x1.contiguous(), x2_gathered.contiguous(), weights, _TriCtx(plan, lib_id, d1, d2, dout))


def contract_segments_differentiable(x1: torch.Tensor, x2: torch.Tensor, weights: torch.Tensor, rowptr: torch.Tensor,
                                     eids: Optional[torch.Tensor], idxs: torch.Tensor, num_atoms: int, scatter_factor: float,
                                     plan: int, lib_id: int, d1: int, d2: int, dout: int) -> torch.Tensor:
    """`Contracter.forward(x1, x2, idxs, N)` (_contract.py:185-211: scale + scatter + gather + contraction) with
    derivatives of every order, on the segmented kernels (see _TriCtx): what training mode runs."""
    t = _TriCtx(plan, lib_id, d1, d2, dout, (rowptr, eids, idxs.reshape(-1), int(num_atoms), float(scatter_factor)))
    return _TriK.apply(x1.contiguous(), x2.contiguous(), weights, t)


# ------------------------------------------------------------------------------------------------------------------
# Training path: linear layers and weighted channels, closed under differentiation
# ------------------------------------------------------------------------------------------------------------------
def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.AA_F32
    if t.dtype == torch.float64:
        return _lib.AA_F64
    raise _lib.AllegroError(f"allegro_amd training ops: dtype {t.dtype}")


@torch.library.custom_op("allegro_amd::linear_wgrad", mutates_args=())
def linear_wgrad(x: torch.Tensor, g: torch.Tensor, lib_id: int) -> torch.Tensor:
    """x [E,K]^T @ g [E,N] -> [K,N] (`aa_linear_wgrad`): the weight gradient of a bias-free linear layer, reduced over the edges
    in fixed-order slabs on the matrix cores."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::linear_wgrad")
    xc = x if x.stride(-1) == 1 and x.dim() == 2 else x.contiguous()
    gc = g if g.stride(-1) == 1 and g.dim() == 2 else g.contiguous()
    E, K, N = xc.shape[0], xc.shape[1], gc.shape[1]
    out = torch.empty((K, N), dtype=x.dtype, device=x.device)
    code = _dtype_code(x)
    nbytes = lib.lib.aa_linear_wgrad_workspace_bytes(code, E, K, N)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    lib.check(lib.lib.aa_linear_wgrad(code, E, K, N, xc.data_ptr() if E else None, xc.stride(0) if E else K, gc.data_ptr() if E else None,
                                      gc.stride(0) if E else N, ws.data_ptr(), nbytes, out.data_ptr(), _stream_ptr(x)), "aa_linear_wgrad")
    return out


@linear_wgrad.register_fake
def _(x, g, lib_id):
    return x.new_empty((x.shape[1], g.shape[1]))


def _rows_ok(w: torch.Tensor) -> torch.Tensor:
    """`w` itself if its rows are contiguous runs (any row stride), a contiguous copy otherwise."""
    return w if (w.dim() == 2 and w.stride(1) == 1 and w.stride(0) >= w.shape[1]) else w.contiguous()


@torch.library.custom_op("allegro_amd::scalar_column", mutates_args=())
def scalar_column_op(a: Optional[torch.Tensor], s: torch.Tensor, D: int, lib_id: int) -> torch.Tensor:
    """`aa_scalar_column`: [E,u,D] = a (None: zeros) with s [E,u] added to component 0."""
    lib = _resolve(lib_id)
    _check_device(lib, s, "allegro_amd::scalar_column")
    sc = _dense16(s)
    ac = None if a is None else _dense16(a)
    E, u = sc.shape
    out = torch.empty((E, u, D), dtype=s.dtype, device=s.device)
    lib.check(lib.lib.aa_scalar_column(_dtype_code(s), E * u, D, ac.data_ptr() if (ac is not None and E) else None, sc.data_ptr() if E else None,
                                       out.data_ptr() if E else None, _stream_ptr(s)), "aa_scalar_column")
    return out


@scalar_column_op.register_fake
def _(a, s, D, lib_id):
    return s.new_empty((s.shape[0], s.shape[1], D))


@torch.library.custom_op("allegro_amd::weighted_channels", mutates_args=())
def weighted_channels_op(which: int, a: torch.Tensor, b: torch.Tensor, u: int, l_max: int, shared: bool, lib_id: int) -> torch.Tensor:
    """`aa_weighted_channels`: which 0: sh [E,D] (x) w [E,u*R] -> [E,u,D];  1: t [E,u,D] . sh -> [E,u*R];  2: t . w -> [E,D]."""
    lib = _resolve(lib_id)
    _check_device(lib, a, "allegro_amd::weighted_channels")
    ac = a.contiguous()
    bc = _rows_ok(b) if which != 1 else b.contiguous()  # (the weight operand keeps its row stride: a column block of an MLP output)
    E = ac.shape[0]
    D, R = (l_max + 1) ** 2, (1 if shared else l_max + 1)
    shape = ((E, u, D), (E, u * R), (E, D))[which]
    out = torch.empty(shape, dtype=a.dtype, device=a.device)
    lib.check(lib.lib.aa_weighted_channels(_dtype_code(a), which, E, u, l_max, int(shared), ac.data_ptr() if E else None,
                                           bc.data_ptr() if E else None, bc.stride(0) if (E and which != 1) else u * R,
                                           out.data_ptr() if E else None, _stream_ptr(a)), "aa_weighted_channels")
    return out


@weighted_channels_op.register_fake
def _(which, a, b, u, l_max, shared, lib_id):
    E, D, R = a.shape[0], (l_max + 1) ** 2, (1 if shared else l_max + 1)
    return a.new_empty(((E, u, D), (E, u * R), (E, D))[which])


@torch.library.custom_op("allegro_amd::weighted_channels_pair", mutates_args=())
def weighted_channels_pair_op(t: torch.Tensor, sh: torch.Tensor, w: torch.Tensor, u: int, l_max: int, shared: bool,
                              lib_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`aa_weighted_channels_pair`: (t . w [E,D], t . sh [E,u*R]) from one pass over t [E,u,D]."""
    lib = _resolve(lib_id)
    _check_device(lib, t, "allegro_amd::weighted_channels_pair")
    tc, sc, wc = t.contiguous(), sh.contiguous(), _rows_ok(w)
    E = tc.shape[0]
    D, R = (l_max + 1) ** 2, (1 if shared else l_max + 1)
    o_sh = torch.empty((E, D), dtype=t.dtype, device=t.device)
    o_w = torch.empty((E, u * R), dtype=t.dtype, device=t.device)
    p = (lambda x: x.data_ptr() if E else None)
    lib.check(lib.lib.aa_weighted_channels_pair(_dtype_code(t), E, u, l_max, int(shared), p(tc), p(sc), p(wc), wc.stride(0) if E else u * R,
                                                p(o_sh), p(o_w), _stream_ptr(t)), "aa_weighted_channels_pair")
    return o_sh, o_w


@weighted_channels_pair_op.register_fake
def _(t, sh, w, u, l_max, shared, lib_id):
    E, D, R = t.shape[0], (l_max + 1) ** 2, (1 if shared else l_max + 1)
    return t.new_empty((E, D)), t.new_empty((E, u * R))


@torch.library.custom_op("allegro_amd::weighted_channels_sum", mutates_args=())
def weighted_channels_sum_op(sh: torch.Tensor, w: torch.Tensor, sh2: torch.Tensor, w2: torch.Tensor, u: int, l_max: int, shared: bool,
                             lib_id: int) -> torch.Tensor:
    """`aa_weighted_channels_sum`: sh (x) w + sh2 (x) w2 -> [E,u,D], one store stream."""
    lib = _resolve(lib_id)
    _check_device(lib, sh, "allegro_amd::weighted_channels_sum")
    a, b, c, d = sh.contiguous(), _rows_ok(w), sh2.contiguous(), _rows_ok(w2)
    E = a.shape[0]
    D, R = (l_max + 1) ** 2, (1 if shared else l_max + 1)
    out = torch.empty((E, u, D), dtype=sh.dtype, device=sh.device)
    p = (lambda x: x.data_ptr() if E else None)
    lib.check(lib.lib.aa_weighted_channels_sum(_dtype_code(sh), E, u, l_max, int(shared), p(a), p(b), b.stride(0) if E else u * R, p(c), p(d),
                                               d.stride(0) if E else u * R, p(out), _stream_ptr(sh)), "aa_weighted_channels_sum")
    return out


@weighted_channels_sum_op.register_fake
def _(sh, w, sh2, w2, u, l_max, shared, lib_id):
    return sh.new_empty((sh.shape[0], u, (l_max + 1) ** 2))


def _dense16(t: torch.Tensor) -> torch.Tensor:
    """Contiguous AND 16-byte aligned: the elementwise kernels use 16-byte vector accesses (AA_REQUIRE on the pointers).  A contiguous
    view with a storage offset -- a row slice, one output of a split -- is contiguous but may start anywhere."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


@torch.library.custom_op("allegro_amd::silu_derivative", mutates_args=())
def silu_derivative_op(x: torch.Tensor, g: Optional[torch.Tensor], order: int, lib_id: int) -> torch.Tensor:
    """`aa_silu_derivative`: g * f^(order)(x), f = SiLU, elementwise (g None: 1)."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::silu_derivative")
    xc = _dense16(x)
    gc = None if g is None else _dense16(g)
    out = torch.empty_like(xc)
    n = xc.numel()
    lib.check(lib.lib.aa_silu_derivative(_dtype_code(x), order, n, xc.data_ptr() if n else None, gc.data_ptr() if (n and gc is not None) else None,
                                         out.data_ptr() if n else None, _stream_ptr(x)), "aa_silu_derivative")
    return out


@silu_derivative_op.register_fake
def _(x, g, order, lib_id):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op("allegro_amd::silu_derivative_pair", mutates_args=())
def silu_derivative_pair_op(x: torch.Tensor, g: torch.Tensor, h: torch.Tensor, order: int, lib_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`aa_silu_derivative_pair`: (g h f^(order+1)(x), h f^(order)(x)) from one pass."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::silu_derivative_pair")
    xc, gc, hc = _dense16(x), _dense16(g), _dense16(h)
    ox, og = torch.empty_like(xc), torch.empty_like(xc)
    n = xc.numel()
    p = (lambda t: t.data_ptr() if n else None)
    lib.check(lib.lib.aa_silu_derivative_pair(_dtype_code(x), order, n, p(xc), p(gc), p(hc), p(ox), p(og), _stream_ptr(x)), "aa_silu_derivative_pair")
    return ox, og


@silu_derivative_pair_op.register_fake
def _(x, g, h, order, lib_id):
    return torch.empty_like(x, memory_format=torch.contiguous_format), torch.empty_like(x, memory_format=torch.contiguous_format)


def _linear_forward_ok(x: torch.Tensor, W: torch.Tensor) -> bool:
    return (x.dtype == torch.float32 and x.dim() == 2 and W.dim() == 2 and x.shape[1] % 32 == 0 and W.shape[1] % 32 == 0 and x.shape[1] >= 32
            and W.shape[1] >= 32 and x.shape[0] > 0)


@torch.library.custom_op("allegro_amd::linear_forward", mutates_args=())
def linear_forward_op(x: torch.Tensor, W: torch.Tensor, lib_id: int) -> torch.Tensor:
    """`aa_linear_forward`: x [E,K] @ W [K,N] (W any strides: a transposed view costs no copy) on the split-bf16 matrix-core kernel."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::linear_forward")
    xc = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
    E, K, N = xc.shape[0], xc.shape[1], W.shape[1]
    out = torch.empty((E, N), dtype=x.dtype, device=x.device)
    nbytes = lib.lib.aa_linear_forward_workspace_bytes(K, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    lib.check(lib.lib.aa_linear_forward(E, K, N, xc.data_ptr(), xc.stride(0), W.data_ptr(), W.stride(0), W.stride(1), ws.data_ptr(), nbytes,
                                        out.data_ptr(), N, _stream_ptr(x)), "aa_linear_forward")
    return out


@linear_forward_op.register_fake
def _(x, W, lib_id):
    return x.new_empty((x.shape[0], W.shape[1]))


def _matmul(x: torch.Tensor, W: torch.Tensor, lib_id: int) -> torch.Tensor:
    """x @ W for a per-edge layer: the hand-written kernel where its shape rules hold, the library GEMM otherwise."""
    if _linear_forward_ok(x, W):
        return torch.ops.allegro_amd.linear_forward(x, W, lib_id)
    return x @ W


@torch.library.custom_op("allegro_amd::act_derivative", mutates_args=())
def act_derivative_op(x: torch.Tensor, g: Optional[torch.Tensor], act: int, order: int, lib_id: int) -> torch.Tensor:
    """`aa_act_derivative`: g * f^(order)(x), f = silu (act 0) / mish (1) / gelu (2), elementwise (g None: 1)."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::act_derivative")
    xc = _dense16(x)
    gc = None if g is None else _dense16(g)
    out = torch.empty_like(xc)
    n = xc.numel()
    lib.check(lib.lib.aa_act_derivative(_dtype_code(x), act, order, n, xc.data_ptr() if n else None,
                                        gc.data_ptr() if (n and gc is not None) else None, out.data_ptr() if n else None, _stream_ptr(x)),
              "aa_act_derivative")
    return out


@act_derivative_op.register_fake
def _(x, g, act, order, lib_id):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op("allegro_amd::act_derivative_pair", mutates_args=())
def act_derivative_pair_op(x: torch.Tensor, g: torch.Tensor, h: torch.Tensor, act: int, order: int, lib_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`aa_act_derivative_pair`: (g h f^(order+1)(x), h f^(order)(x)) from one pass."""
    lib = _resolve(lib_id)
    _check_device(lib, x, "allegro_amd::act_derivative_pair")
    xc, gc, hc = _dense16(x), _dense16(g), _dense16(h)
    ox, og = torch.empty_like(xc), torch.empty_like(xc)
    n = xc.numel()
    p = (lambda t: t.data_ptr() if n else None)
    lib.check(lib.lib.aa_act_derivative_pair(_dtype_code(x), act, order, n, p(xc), p(gc), p(hc), p(ox), p(og), _stream_ptr(x)), "aa_act_derivative_pair")
    return ox, og


@act_derivative_pair_op.register_fake
def _(x, g, h, act, order, lib_id):
    return torch.empty_like(x, memory_format=torch.contiguous_format), torch.empty_like(x, memory_format=torch.contiguous_format)


class _Act(torch.autograd.Function):
    """`_Silu` for any of the reference's nonlinearities: A_k(x, g) = g f^(k)(x), f = silu / mish / gelu (`act` 0 / 1 / 2)."""

    @staticmethod
    def forward(ctx, x, g, k, act, lib_id):
        ctx.k, ctx.act, ctx.lib_id, ctx.has_g = k, act, lib_id, g is not None
        ctx.save_for_backward(x, g)
        return torch.ops.allegro_amd.act_derivative(x.detach(), None if g is None else g.detach(), act, k, lib_id)

    @staticmethod
    def backward(ctx, h):
        x, g = ctx.saved_tensors
        k, act, lib_id = ctx.k, ctx.act, ctx.lib_id
        need_x, need_g = ctx.needs_input_grad[0], ctx.has_g and ctx.needs_input_grad[1]
        if k >= 3 and need_x:
            raise NotImplementedError("allegro_amd: activation derivatives beyond the third are not implemented")
        if need_x and need_g and not torch.is_grad_enabled():
            gx, gg = torch.ops.allegro_amd.act_derivative_pair(x, g, h, act, k, lib_id)
            return gx, gg, None, None, None
        gx = _Act.apply(x, h if g is None else g * h, k + 1, act, lib_id) if need_x else None
        gg = _Act.apply(x, h, k, act, lib_id) if need_g else None
        return gx, gg, None, None, None


ACT_CODES = {"silu": 0, "mish": 1, "gelu": 2}


def activation(x: torch.Tensor, kind: str, lib_id: int) -> torch.Tensor:
    """The hidden nonlinearity of a scalar MLP in training mode (silu / mish / gelu), differentiable to third order, on the
    elementwise family kernel (`aa_act_derivative`)."""
    return _Silu.apply(x, None, 0, lib_id) if kind == "silu" else _Act.apply(x, None, 0, ACT_CODES[kind], lib_id)


class _Silu(torch.autograd.Function):
    """A_k(x, g) = g f^(k)(x) with f = SiLU (g None: f^(k)(x)).  d/dx = A_{k+1}(x, g .), d/dg = A_k(x, .): closed under
    differentiation, one launch per member; where no further derivative is recorded (the backward pass of the loss) both gradients
    come from one pass (`silu_derivative_pair`)."""

    @staticmethod
    def forward(ctx, x, g, k, lib_id):
        ctx.k, ctx.lib_id, ctx.has_g = k, lib_id, g is not None
        ctx.save_for_backward(x, g)
        return torch.ops.allegro_amd.silu_derivative(x.detach(), None if g is None else g.detach(), k, lib_id)

    @staticmethod
    def backward(ctx, h):
        x, g = ctx.saved_tensors
        k, lib_id = ctx.k, ctx.lib_id
        need_x, need_g = ctx.needs_input_grad[0], ctx.has_g and ctx.needs_input_grad[1]
        if k >= 3 and need_x:
            raise NotImplementedError("allegro_amd: SiLU derivatives beyond the third are not implemented")
        if need_x and need_g and not torch.is_grad_enabled():
            gx, gg = torch.ops.allegro_amd.silu_derivative_pair(x, g, h, k, lib_id)
            return gx, gg, None, None
        gx = _Silu.apply(x, h if g is None else g * h, k + 1, lib_id) if need_x else None
        gg = _Silu.apply(x, h, k, lib_id) if need_g else None
        return gx, gg, None, None


def silu(x: torch.Tensor, lib_id: int) -> torch.Tensor:
    """SiLU of a hidden layer in training mode, differentiable to third order, on the elementwise family kernel."""
    return _Silu.apply(x, None, 0, lib_id)


class _MM(torch.autograd.Function):
    """y = x @ W for a per-edge linear layer (x [E,K], W [K,N]).  The products along the edges are library GEMMs; the one
    product that reduces OVER the edges -- the weight gradient -- is `_XtG` (hand-written).  With `_XtG` the pair is closed
    under differentiation, so forces (first derivatives) can be differentiated again by a force-matching loss."""

    @staticmethod
    def forward(ctx, x, W, lib_id):
        ctx.lib_id = lib_id
        ctx.save_for_backward(x, W)
        return _matmul(x.detach(), W.detach(), lib_id)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        n = ctx.needs_input_grad
        return (_MM.apply(g, W.t(), ctx.lib_id) if n[0] else None, _XtG.apply(x, g, ctx.lib_id) if n[1] else None, None)


class _XtG(torch.autograd.Function):
    """x [E,K]^T @ g [E,N]."""

    @staticmethod
    def forward(ctx, x, g, lib_id):
        ctx.lib_id = lib_id
        ctx.save_for_backward(x, g)
        return torch.ops.allegro_amd.linear_wgrad(x.detach(), g.detach(), lib_id)

    @staticmethod
    def backward(ctx, G):  # G [K,N]
        x, g = ctx.saved_tensors
        n = ctx.needs_input_grad
        return (_MM.apply(g, G.t(), ctx.lib_id) if n[0] else None, _MM.apply(x, G, ctx.lib_id) if n[1] else None, None)


def linear(x: torch.Tensor, W: torch.Tensor, lib_id: int) -> torch.Tensor:
    """Bias-free linear layer of a ScalarMLPFunction in training mode, differentiable to any order."""
    return _MM.apply(x, W, lib_id)


class _WcB(torch.autograd.Function):
    """B(sh, w)[e,c,i] = sh[e,i] w[e,c,r(i)]."""

    @staticmethod
    def forward(ctx, sh, w, meta):
        ctx.meta = meta
        ctx.save_for_backward(sh, w)
        return torch.ops.allegro_amd.weighted_channels(0, sh.detach(), w.detach(), *meta)

    @staticmethod
    def backward(ctx, g):
        sh, w = ctx.saved_tensors
        n = ctx.needs_input_grad
        if n[0] and n[1]:  # (the usual case: both operands depend on the positions) one pass over g for both
            return _WcPair.apply(g, sh, w, ctx.meta) + (None,)
        return (_WcS.apply(g, w, ctx.meta) if n[0] else None, _WcW.apply(g, sh, ctx.meta) if n[1] else None, None)


class _WcS(torch.autograd.Function):
    """[e,i] = sum_c t[e,c,i] w[e,c,r(i)]  (the sh gradient)."""

    @staticmethod
    def forward(ctx, t, w, meta):
        ctx.meta = meta
        ctx.save_for_backward(t, w)
        return torch.ops.allegro_amd.weighted_channels(2, t.detach(), w.detach(), *meta)

    @staticmethod
    def backward(ctx, h):
        t, w = ctx.saved_tensors
        n = ctx.needs_input_grad
        return (_WcB.apply(h, w, ctx.meta) if n[0] else None, _WcW.apply(t, h, ctx.meta) if n[1] else None, None)


class _WcW(torch.autograd.Function):
    """[e,c,r] = sum_{i in r} t[e,c,i] sh[e,i]  (the weight gradient)."""

    @staticmethod
    def forward(ctx, t, sh, meta):
        ctx.meta = meta
        ctx.save_for_backward(t, sh)
        return torch.ops.allegro_amd.weighted_channels(1, t.detach(), sh.detach(), *meta)

    @staticmethod
    def backward(ctx, q):
        t, sh = ctx.saved_tensors
        n = ctx.needs_input_grad
        return (_WcB.apply(sh, q, ctx.meta) if n[0] else None, _WcS.apply(t, q, ctx.meta) if n[1] else None, None)


class _WcPair(torch.autograd.Function):
    """(S(t, w), W(t, sh)) = (sum_c t w, sum_{i in r} t sh) from ONE pass over t [E,u,D] -- the two gradients of B(sh, w).  Its own
    derivative is the same pair with other second operands plus the two-term B: the family stays closed."""

    @staticmethod
    def forward(ctx, t, sh, w, meta):
        ctx.meta = meta
        ctx.set_materialize_grads(False)  # (an unused output arrives as None, not as a zero tensor to be multiplied through)
        ctx.save_for_backward(t, sh, w)
        return torch.ops.allegro_amd.weighted_channels_pair(t.detach(), sh.detach(), w.detach(), *meta)

    @staticmethod
    def backward(ctx, a_s, a_w):  # cotangents of S [E,D] and of W [E,u*R]
        t, sh, w = ctx.saved_tensors
        n = ctx.needs_input_grad
        meta = ctx.meta
        gt = gsh = gw = None
        if a_s is None and a_w is None:
            return None, None, None, None
        if n[0]:  # d/dt: B(a_s, w) + B(sh, a_w)
            if a_s is not None and a_w is not None:
                gt = _WcB2.apply(a_s, w, sh, a_w, meta)
            elif a_s is not None:
                gt = _WcB.apply(a_s, w, meta)
            elif a_w is not None:
                gt = _WcB.apply(sh, a_w, meta)
        # d/dsh: S(t, a_w) (through W);  d/dw: W(t, a_s) (through S)
        if n[1] and n[2] and a_s is not None and a_w is not None:
            gsh, gw = _WcPair.apply(t, a_s, a_w, meta)
        else:
            if n[1] and a_w is not None:
                gsh = _WcS.apply(t, a_w, meta)
            if n[2] and a_s is not None:
                gw = _WcW.apply(t, a_s, meta)
        return gt, gsh, gw, None


class _WcB2(torch.autograd.Function):
    """B(sh, w) + B(sh2, w2) with one store stream."""

    @staticmethod
    def forward(ctx, sh, w, sh2, w2, meta):
        ctx.meta = meta
        ctx.save_for_backward(sh, w, sh2, w2)
        return torch.ops.allegro_amd.weighted_channels_sum(sh.detach(), w.detach(), sh2.detach(), w2.detach(), *meta)

    @staticmethod
    def backward(ctx, g):
        sh, w, sh2, w2 = ctx.saved_tensors
        n = ctx.needs_input_grad
        meta = ctx.meta

        def pair(need_s, need_w, a, b):
            if need_s and need_w:
                return _WcPair.apply(g, a, b, meta)
            return (_WcS.apply(g, b, meta) if need_s else None, _WcW.apply(g, a, meta) if need_w else None)

        return pair(n[0], n[1], sh, w) + pair(n[2], n[3], sh2, w2) + (None,)


def weighted_channels(sh: torch.Tensor, w: torch.Tensor, u: int, l_max: int, lib_id: int) -> torch.Tensor:
    """MakeWeightedChannels (allegro/nn/_strided/_channels.py:44-63) -> [E,u,D], differentiable to any order, one kernel pass per
    evaluation.  `w` [E, u (l_max+1)] (one weight per channel and irrep) or [E, u] (`weight_individual_irreps=False`)."""
    shared = w.shape[1] == u and l_max > 0
    return _WcB.apply(sh.contiguous(), _rows_ok(w), (int(u), int(l_max), bool(shared), int(lib_id)))


class _TakeCol0(torch.autograd.Function):
    """t[:, :, 0] of a tensor feature [E,u,D] as a contiguous [E,u]; its transpose is the zero-padded form of `_AddCol0`."""

    @staticmethod
    def forward(ctx, t, lib_id):
        ctx.D, ctx.lib_id = t.shape[2], lib_id
        return t[:, :, 0].contiguous()

    @staticmethod
    def backward(ctx, g):
        return _AddCol0.apply(None, g, ctx.D, ctx.lib_id), None


class _AddCol0(torch.autograd.Function):
    """a [E,u,D] (None: zeros) with s [E,u] added to component 0 -- one pass (`aa_scalar_column`)."""

    @staticmethod
    def forward(ctx, a, s, D, lib_id):
        ctx.has_a, ctx.lib_id = a is not None, lib_id
        return torch.ops.allegro_amd.scalar_column(None if a is None else a.detach(), s.detach(), D, lib_id)

    @staticmethod
    def backward(ctx, h):
        n = ctx.needs_input_grad
        return (h if (ctx.has_a and n[0]) else None, _TakeCol0.apply(h, ctx.lib_id) if n[1] else None, None, None)


class _Fork0(torch.autograd.Function):
    """(t, t[:, :, 0]) as ONE node: the tensor feature goes on to the next tensor product, its scalars to the next latent MLP
    (_allegro.py:275-283); the two gradients meet here and are merged in one pass instead of autograd's zero fill + strided copy +
    addition of two [E,u,D] buffers."""

    @staticmethod
    def forward(ctx, t, lib_id):
        ctx.D, ctx.lib_id = t.shape[2], lib_id
        ctx.set_materialize_grads(False)
        return t.view_as(t), t[:, :, 0].contiguous()

    @staticmethod
    def backward(ctx, g_t, g_s):
        if g_s is None:
            return g_t, None
        return _AddCol0.apply(g_t, g_s, ctx.D, ctx.lib_id), None


def fork_scalars(t: torch.Tensor, lib_id: int):
    """(t, t[:, :, 0]) of a tensor feature [E,u,D], differentiable to any order."""
    return _Fork0.apply(t, lib_id)


@torch.library.custom_op("allegro_amd::concat_columns", mutates_args=())
def concat_columns_op(xs: List[torch.Tensor], lib_id: int) -> torch.Tensor:
    """`aa_concat_columns`: [E, sum of widths] from up to 8 [E, w_j] tensors (row-strided views taken as they are)."""
    import ctypes as C

    lib = _resolve(lib_id)
    _check_device(lib, xs[0], "allegro_amd::concat_columns")
    xc = [_rows_ok(x) for x in xs]
    E, n = xc[0].shape[0], len(xc)
    total = sum(int(x.shape[1]) for x in xc)
    out = torch.empty((E, total), dtype=xc[0].dtype, device=xc[0].device)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() if E else None for x in xc])
    lds = (C.c_int64 * n)(*[x.stride(0) if E else x.shape[1] for x in xc])
    ws = (C.c_int * n)(*[int(x.shape[1]) for x in xc])
    lib.check(lib.lib.aa_concat_columns(_dtype_code(xc[0]), E, n, ptrs, lds, ws, out.data_ptr() if E else None, total, _stream_ptr(xc[0])),
              "aa_concat_columns")
    return out


@concat_columns_op.register_fake
def _(xs, lib_id):
    return xs[0].new_empty((xs[0].shape[0], sum(int(x.shape[1]) for x in xs)))


def _concat(xs, lib_id):
    if lib_id is None or len(xs) > 8 or any(x.dim() != 2 or x.shape[1] == 0 for x in xs):
        return torch.cat(list(xs), dim=1)
    return torch.ops.allegro_amd.concat_columns(list(xs), lib_id)


class _Split(torch.autograd.Function):
    """Column blocks of x [E, sum] as views; the gradient is one concatenation (`_Cat`)."""

    @staticmethod
    def forward(ctx, x, sizes, lib_id):
        ctx.lib_id = lib_id
        return tuple(v.view_as(v) for v in torch.split(x, sizes, dim=1))

    @staticmethod
    def backward(ctx, *gs):
        return _Cat.apply(ctx.lib_id, *gs), None, None


def split_columns(x: torch.Tensor, sizes, lib_id: int):
    return _Split.apply(x, [int(v) for v in sizes], lib_id)


class _Cat(torch.autograd.Function):
    """Concatenation along the feature axis whose gradient is ONE `split` node (its own gradient: one concatenation).  `torch.cat`'s
    gradient is a `narrow` per input, and each of those, differentiated again by a force loss, is a zero-filled full-width buffer,
    a strided copy and an addition (1.2 ms per training step at C3, profiles/r05_v42_train_nodes.txt)."""

    @staticmethod
    def forward(ctx, lib_id, *xs):
        ctx.sizes, ctx.lib_id = [int(x.shape[1]) for x in xs], lib_id
        return _concat([x.detach() for x in xs], lib_id)

    @staticmethod
    def backward(ctx, g):
        return (None,) + _Split.apply(g, ctx.sizes, ctx.lib_id)


def cat_features(xs, lib_id: Optional[int] = None) -> torch.Tensor:
    return _Cat.apply(lib_id, *xs)


class _EdgeDiff(torch.autograd.Function):
    """vec[e] = x[nbr[e]] - x[center[e]] for per-atom rows x [N,c].  Its transpose is two deterministic segment sums (own CSR for
    the centers, transposed CSR for the neighbours) instead of two atomic `index_add`s of E rows (0.65 ms each at C3,
    profiles/r05_v9_train_c3_kernel_stats_first_kernels.txt); the pair (_EdgeDiff, _EdgeDiffT) is closed under differentiation."""

    @staticmethod
    def forward(ctx, x, csr):
        ctx.csr = csr
        center, nbr = csr[0], csr[1]
        return x.index_select(0, nbr) - x.index_select(0, center)

    @staticmethod
    def backward(ctx, g):
        return _EdgeDiffT.apply(g, ctx.csr), None


class _EdgeDiffT(torch.autograd.Function):
    """out[a] = sum_{e: nbr[e] = a} g[e] - sum_{e: center[e] = a} g[e]."""

    @staticmethod
    def forward(ctx, g, csr):
        ctx.csr = csr
        _center, _nbr, rowptr, t_rowptr, t_perm, n, lib_id = csr
        gd = g.detach()
        own = torch.ops.allegro_amd.segment_sum(gd, rowptr, None, n, 1.0, lib_id)
        other = torch.ops.allegro_amd.segment_sum(gd, t_rowptr, t_perm, n, 1.0, lib_id)
        return other - own

    @staticmethod
    def backward(ctx, h):
        return _EdgeDiff.apply(h, ctx.csr), None


def edge_difference(x: torch.Tensor, graph, lib_id: int) -> torch.Tensor:
    """`with_edge_vectors_` without the shift (called at allegro/nn/tensorembed.py:86): x[nbr] - x[center] on a center-sorted
    `PreparedGraph` with its transposed CSR, differentiable to any order with segment-sum transposes."""
    csr = (graph.center.long(), graph.nbr.long(), graph.rowptr, graph.t_rowptr, graph.t_perm, int(graph.num_atoms), int(lib_id))
    return _EdgeDiff.apply(x, csr)


class _SegSum(torch.autograd.Function):
    """out[a] = sum over the edges of center a (EdgewiseReduce, allegro/nn/edgewise.py:40-60) on the segment-sum kernel; its
    transpose is a gather."""

    @staticmethod
    def forward(ctx, x, center, rowptr, n, lib_id):
        ctx.center, ctx.rowptr, ctx.n, ctx.lib_id = center, rowptr, n, lib_id
        return torch.ops.allegro_amd.segment_sum(x.detach(), rowptr, None, n, 1.0, lib_id)

    @staticmethod
    def backward(ctx, g):
        return _Gather.apply(g, ctx.center, ctx.rowptr, ctx.n, ctx.lib_id), None, None, None, None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, center, rowptr, n, lib_id):
        ctx.center, ctx.rowptr, ctx.n, ctx.lib_id = center, rowptr, n, lib_id
        return x.index_select(0, center)

    @staticmethod
    def backward(ctx, g):
        return _SegSum.apply(g, ctx.center, ctx.rowptr, ctx.n, ctx.lib_id), None, None, None, None


def edge_to_atom_sum(x: torch.Tensor, graph, lib_id: int) -> torch.Tensor:
    return _SegSum.apply(x.contiguous(), graph.center.long(), graph.rowptr, int(graph.num_atoms), int(lib_id))
