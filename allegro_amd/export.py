"""Export path: the whole step as ONE dispatcher op that `torch.export` / AOTInductor can capture and a Python-free
host can call.

The reference ships models to LAMMPS by `nequip-compile`-ing them into an AOTInductor package whose tensor contract
is `[pos, edge_index, atom_types] -> LMP_OUTPUTS` with ghost atoms appended (allegro/_compile.py:10-14,17-65).
`ExportableAllegro` honours that contract around `allegro_amd_native::energy_forces`
(allegro_amd/csrc/torch_ops.cpp, registered from C++ so that it exists in any process that loads
`liballegro_amd_torch.so`): hyper-parameters + Clebsch-Gordan tables travel as an int list, the packed weights
as a device byte tensor -- both become constants of the exported program.
"""
import ctypes as C
import struct
from typing import Optional

import torch

from . import _lib
from .build import build_torch_ops

_LOADED = False
_MAGIC = 0x414C4C4547524F32  # "ALLEGRO2": config format 2 (torch_ops.cpp rejects packages of other formats: their blob layout differs)


def load_native_ops() -> None:
    """Loads `liballegro_amd_torch.so` (built in-tree by allegro_amd.build) into the dispatcher."""
    global _LOADED
    if not _LOADED:
        torch.ops.load_library(build_torch_ops(verbose=False))
        _LOADED = True


def _bits(x: float) -> int:
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def serialize_config(model, layout_hash: int = 0) -> list:
    """`aa_model_config` of a HipAllegroModel as the int64 word list torch_ops.cpp parses; `layout_hash` is the digest of
    the blob layout the weights were packed for (aa_model_plan_layout_hash; mandatory: the op refuses 0)."""
    cfg, _keep = model._plan_keep if getattr(model, "_plan_keep", None) else model._build_config()  # (no library call: works without a GPU)
    w = [_MAGIC, cfg.dtype, cfg.num_types, cfg.num_bessels, cfg.l_max, cfg.num_layers, cfg.num_scalar, cfg.num_tensor,
         cfg.embed_dim, cfg.embed_mlp_depth, cfg.embed_mlp_width, cfg.latent_mlp_depth, cfg.latent_mlp_width,
         cfg.readout_mlp_depth, cfg.readout_mlp_width, cfg.forward_weight_init, cfg.has_scales, cfg.has_shifts,
         cfg.embed_kind, cfg.spline_span, _bits(cfg.poly_p), _bits(cfg.avg_num_neighbors), _bits(cfg.act_const),
         cfg.env_shared_weights, cfg.act_kind[0] | (cfg.act_kind[1] << 8) | (cfg.act_kind[2] << 16),
         _bits(cfg.act_consts[0]), _bits(cfg.act_consts[1]), _bits(cfg.act_consts[2]), cfg.bessel_convention,
         layout_hash - (1 << 64) if layout_hash >= (1 << 63) else layout_hash]
    for l in range(cfg.num_layers):
        d = cfg.tps[l]
        w += [d.mul, d.d1, d.d2, d.dout, d.num_paths, d.coupling, d.nnz]
        for arr in (d.nz_i, d.nz_j, d.nz_k, d.nz_path):
            w += [int(arr[t]) for t in range(d.nnz)]
        w += [_bits(d.nz_val[t]) for t in range(d.nnz)]
    return [int(v) for v in w]


def write_host_model(model, path: str) -> None:
    """Model file for Python-free hosts (`aa_model_file_open`, include/allegro_amd.h section 5; layout in
    csrc/aa_hostfile.hip): the serialized `aa_model_config` (hyper-parameters + Clebsch-Gordan non-zeros) followed by every
    parameter as float64 in the reference's own state_dict layout.  Needs no GPU.  A C host then runs
    `aa_model_plan_create(aa_model_file_config(f))` / `aa_model_pack_weights(plan, aa_model_file_weights(f), ...)`
    (tests/host/host_c99.c)."""
    import numpy as np

    words = np.asarray(serialize_config(model, 0), dtype="<i8")
    tensors = model._raw_tensors()
    with open(path, "wb") as f:
        f.write(b"AAMODEL1")
        f.write(np.asarray([words.size], dtype="<i8").tobytes())
        f.write(words.tobytes())
        f.write(np.asarray([len(tensors)], dtype="<i8").tobytes())
        for slot, t in tensors:
            a = np.ascontiguousarray(t.detach().cpu().double().numpy().reshape(-1)).astype("<f8")
            f.write(np.asarray([slot, a.size], dtype="<i8").tobytes())
            f.write(a.tobytes())


class ExportableAllegro(torch.nn.Module):
    """`forward(pos, edge_index, atom_types[, shift_vec]) -> (atomic_energy [N,1], total_energy [1,1], forces [N,3],
    virial [1,3,3])` -- the `LMP_OUTPUTS` of the reference's `pair_allegro` compile target (allegro/_compile.py:68-74;
    the key list itself is nequip's, EXT: per-atom energy, total energy, forces, virial) -- through the
    C++-registered op; build it from a `HipAllegroModel` whose weights are final.  virial = -dE/d(strain)
    (nequip ForceStressOutput convention, which is also LAMMPS' sign); in the ghost-atom layout it is the sum over
    the edges handed in, i.e. the local atoms' share, like every other output."""

    def __init__(self, model, device):
        super().__init__()
        load_native_ops()
        device = torch.device(device)
        model._select_device(device)
        model._ensure_plan()
        # The op creates its plan with DEFAULT aa_plan_options, whatever AA_* switches the Python host of this process
        # maps onto the model's own plan, and several of those switches change the blob layout (column order, padding,
        # moments split): pack the blob for a default-options plan, and store that layout's digest in the config so that
        # the op can refuse a blob packed for another layout instead of computing garbage from it.
        lib = model._get_lib()
        cfg, _keep = model._plan_keep
        plan = lib.model_plan_create(cfg, _lib.PlanOptions())
        try:
            blob = model._pack_blob(plan, device)
            layout = int(lib.lib.aa_model_plan_layout_hash(plan))
        finally:
            lib.model_plan_destroy(plan)
        self.config = serialize_config(model, layout)  # int list (a constant of the exported program)
        self.register_buffer("weights", blob)

    def forward(self, pos: torch.Tensor, edge_index: torch.Tensor, atom_types: torch.Tensor,
                shift_vec: Optional[torch.Tensor] = None):
        e_atom, forces, virial = torch.ops.allegro_amd_native.energy_forces(pos, edge_index, atom_types, shift_vec,
                                                                            self.config, self.weights)
        return e_atom.unsqueeze(-1), e_atom.sum().reshape(1, 1), forces, virial
