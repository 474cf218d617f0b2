"""ORACLE SHIM: the `nequip.nn` leaves the reference's hot path calls.

Restated from memory of nequip >=0.13 -- PARITY UNPINNED against nequip itself.
Reference call sites are cited per symbol.
"""
import math
from collections import OrderedDict
from typing import Dict, Optional

import torch

from e3nn.o3._irreps import Irrep, Irreps

from ..data import AtomicDataDict
from ..data.AtomicDataDict import with_edge_vectors_  # noqa: F401  (tensorembed.py:86)


# --------------------------------------------------------------------------- scatter
def scatter(src, index, dim: int = 0, dim_size=None, reduce: str = "sum"):
    """zeros + index_add_.  Call sites: _contract.py:199-204, edgewise.py:52-58."""
    assert dim == 0 and reduce == "sum"
    if dim_size is None:
        dim_size = int(index.max()) + 1
    if isinstance(dim_size, torch.Tensor):
        dim_size = int(dim_size.reshape(-1)[0])
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    """Call site: _allegro.py:126."""
    ir_out = Irrep(ir_out)
    for _, ir1 in Irreps(irreps_in1):
        for _, ir2 in Irreps(irreps_in2):
            if ir_out in ir1 * ir2:
                return True
    return False


# --------------------------------------------------------------------------- plugin plumbing
def model_modifier(persistent: bool):
    def deco(fn):
        try:
            fn._is_model_modifier = True
            fn._persistent = persistent
        except AttributeError:
            pass
        return fn

    return deco


def replace_submodules(model, target_cls, factory):
    """Recursively replace every instance of `target_cls` by `factory(old)` (_contract.py:282,310)."""
    if isinstance(model, target_cls):
        return factory(model)
    for name, child in list(model.named_children()):
        setattr(model, name, replace_submodules(child, target_cls, factory))
    return model


# --------------------------------------------------------------------------- graph module base
class GraphModuleMixin:
    def _init_irreps(self, irreps_in=None, my_irreps_in=None, required_irreps_in=(), irreps_out=None):
        irreps_in = {} if irreps_in is None else dict(irreps_in)
        irreps_in = {k: (None if v is None else Irreps(v)) for k, v in irreps_in.items()}
        for k in required_irreps_in:
            assert k in irreps_in, f"missing required irreps_in[{k}]"
        self.irreps_in = irreps_in
        new_out = dict(irreps_in)
        if irreps_out:
            new_out.update({k: (None if v is None else Irreps(v)) for k, v in irreps_out.items()})
        self.irreps_out = new_out


class SequentialGraphNetwork(GraphModuleMixin, torch.nn.Sequential):
    def __init__(self, modules: Dict[str, torch.nn.Module]):
        mods = list(modules.values())
        super().__init__(OrderedDict(modules))
        self._init_irreps(irreps_in=mods[0].irreps_in, irreps_out=mods[-1].irreps_out)

    def forward(self, data):
        for m in self:
            data = m(data)
        return data


# --------------------------------------------------------------------------- scalar MLP
def _second_moment_const(act) -> float:
    """1/sqrt(E_{z~N(0,1)}[act(z)^2]) (e3nn `normalize2mom`), by deterministic quadrature."""
    z = torch.linspace(-12.0, 12.0, 240001, dtype=torch.float64)
    w = torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    m2 = torch.trapezoid(act(z) ** 2 * w, z).item()
    return 1.0 / math.sqrt(m2)


class ScalarLinearLayer(torch.nn.Module):
    def __init__(self, in_features: int, out_features: int, alpha: float, bias: bool):
        super().__init__()
        self.alpha = alpha
        self.weight = torch.nn.Parameter(torch.empty(in_features, out_features))
        torch.nn.init.uniform_(self.weight, -math.sqrt(3), math.sqrt(3))
        self.bias = torch.nn.Parameter(torch.zeros(out_features)) if bias else None

    def forward(self, x):
        y = torch.mm(x, self.weight * self.alpha)
        if self.bias is not None:
            y = y + self.bias
        return y


class ScalarMLPFunction(torch.nn.Module):
    """dims [in]+depth*[width]+[out]; y = x @ (W_i*alpha_i), W_i:[in,out]~U(-sqrt3,sqrt3),
    alpha_i = c_prev/sqrt(fan_in|fan_out), c_prev = normalize2mom const after an activation.
    Call sites: _allegro.py:90-94,193-213; tensorembed.py:76-81; _edgeembed.py:59-64."""

    def __init__(self, input_dim, output_dim, hidden_layers_depth: int = 0, hidden_layers_width=None,
                 nonlinearity: Optional[str] = "silu", bias: bool = False, forward_weight_init: bool = True):
        super().__init__()
        act = {None: None, "silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu,
               "mish": torch.nn.functional.mish}[nonlinearity]
        self.dims = [input_dim] + hidden_layers_depth * [hidden_layers_width] + [output_dim]
        self.num_layers = len(self.dims) - 1
        self.is_nonlinear = hidden_layers_depth > 0 and act is not None
        self.nonlinearity = nonlinearity
        self.act_const = _second_moment_const(act) if act is not None else 1.0
        layers = OrderedDict()
        norm_from_last = 1.0
        for i, (h_in, h_out) in enumerate(zip(self.dims, self.dims[1:])):
            alpha = norm_from_last / math.sqrt(float(h_in if forward_weight_init else h_out))
            layers[str(i)] = ScalarLinearLayer(h_in, h_out, alpha, bias)
            if i < self.num_layers - 1 and act is not None:
                layers[f"activation_{i}"] = {"silu": torch.nn.SiLU, "gelu": torch.nn.GELU, "mish": torch.nn.Mish}[nonlinearity]()
                norm_from_last = self.act_const
        self.mlp = torch.nn.Sequential(layers)

    def forward(self, x):
        return self.mlp(x)


class ScalarMLP(GraphModuleMixin, torch.nn.Module):
    """GraphModule wrapper (allegro_models.py:173-183,231-241)."""

    def __init__(self, output_dim, hidden_layers_depth=0, hidden_layers_width=None, nonlinearity="silu",
                 bias=False, forward_weight_init=True, field=None, out_field=None, irreps_in=None):
        super().__init__()
        self.field = field
        self.out_field = out_field if out_field is not None else field
        self._init_irreps(irreps_in=irreps_in, required_irreps_in=[field],
                          irreps_out={self.out_field: Irreps([(output_dim, (0, 1))])})
        self.mlp = ScalarMLPFunction(
            input_dim=self.irreps_in[field].num_irreps, output_dim=output_dim,
            hidden_layers_depth=hidden_layers_depth, hidden_layers_width=hidden_layers_width,
            nonlinearity=nonlinearity, bias=bias, forward_weight_init=forward_weight_init)

    def forward(self, data):
        data[self.out_field] = self.mlp(data[self.field])
        return data


# --------------------------------------------------------------------------- downstream modules
class PerTypeScaleShift(GraphModuleMixin, torch.nn.Module):
    def __init__(self, type_names, field, out_field, scales=None, shifts=None, scales_trainable=False,
                 shifts_trainable=False, irreps_in=None):
        super().__init__()
        self.field, self.out_field = field, out_field
        self._init_irreps(irreps_in=irreps_in, irreps_out={out_field: irreps_in[field]})
        nt = len(type_names)

        def prep(v):
            if v is None:
                return None
            if isinstance(v, dict):
                v = [v[t] for t in type_names]
            t = torch.as_tensor(v, dtype=torch.get_default_dtype()).reshape(-1)
            return t.expand(nt).clone() if t.numel() == 1 else t

        s, b = prep(scales), prep(shifts)
        self.has_scales, self.has_shifts = s is not None, b is not None
        if s is not None:
            self.scales = torch.nn.Parameter(s, requires_grad=bool(scales_trainable))
        if b is not None:
            self.shifts = torch.nn.Parameter(b, requires_grad=bool(shifts_trainable))

    def forward(self, data):
        x = data[self.field]
        t = data[AtomicDataDict.ATOM_TYPE_KEY].reshape(-1)
        if self.has_scales:
            x = x * self.scales[t].reshape(-1, 1)
        if self.has_shifts:
            x = x + self.shifts[t].reshape(-1, 1)
        data[self.out_field] = x
        return data


class AtomwiseReduce(GraphModuleMixin, torch.nn.Module):
    def __init__(self, field, out_field=None, reduce="sum", irreps_in=None):
        super().__init__()
        assert reduce == "sum"
        self.field, self.out_field = field, out_field
        self._init_irreps(irreps_in=irreps_in, irreps_out={out_field: irreps_in[field]})

    def forward(self, data):
        x = data[self.field]
        if AtomicDataDict.BATCH_KEY in data:
            nf = AtomicDataDict.num_frames(data)
            data[self.out_field] = scatter(x, data[AtomicDataDict.BATCH_KEY], 0, nf)
        else:
            data[self.out_field] = x.sum(dim=0, keepdim=True)
        return data


class ForceStressOutput(GraphModuleMixin, torch.nn.Module):
    """forces = -d(total_energy)/d(pos) by autograd (allegro_models.py:103,305). Stress not restated."""

    def __init__(self, func):
        super().__init__()
        self.func = func
        self._init_irreps(irreps_in=func.irreps_in, irreps_out=func.irreps_out)

    def forward(self, data):
        pos = data[AtomicDataDict.POSITIONS_KEY]
        if not pos.requires_grad:
            pos = pos.detach().clone().requires_grad_(True)
        data[AtomicDataDict.POSITIONS_KEY] = pos
        data = self.func(data)
        (g,) = torch.autograd.grad([data[AtomicDataDict.TOTAL_ENERGY_KEY].sum()], [pos], create_graph=self.training)
        data[AtomicDataDict.FORCE_KEY] = -g
        return data
