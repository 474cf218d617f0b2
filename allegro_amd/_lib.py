"""ctypes binding of the C ABI in include/allegro_amd.h.

`load()` returns the gfx950 library (building it in-tree with hipcc when the sources are newer) and
raises if it cannot -- there is NO CPU fallback in the product path.  `AllegroLib` is parameterised
by the CDLL handle only so that tests can bind the same wrapper to their emulation build.
"""
import ctypes as C
import os
from typing import Optional

import numpy as np

AA_MAX_LAYERS = 4
AA_MAX_MLP_LAYERS = 4
AA_F32, AA_F64 = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class TpDesc(C.Structure):
    _fields_ = [("mul", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32), ("dout", C.c_int32),
                ("num_paths", C.c_int32), ("coupling", C.c_int32), ("nnz", C.c_int32),
                ("nz_i", _ip), ("nz_j", _ip), ("nz_k", _ip), ("nz_path", _ip), ("nz_val", _dp)]


class ModelConfig(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("num_types", C.c_int32), ("num_bessels", C.c_int32), ("poly_p", C.c_double),
                ("l_max", C.c_int32), ("num_layers", C.c_int32), ("num_scalar", C.c_int32), ("num_tensor", C.c_int32),
                ("embed_dim", C.c_int32), ("embed_mlp_depth", C.c_int32), ("embed_mlp_width", C.c_int32),
                ("latent_mlp_depth", C.c_int32), ("latent_mlp_width", C.c_int32),
                ("readout_mlp_depth", C.c_int32), ("readout_mlp_width", C.c_int32),
                ("forward_weight_init", C.c_int32), ("avg_num_neighbors", C.c_double), ("act_const", C.c_double),
                ("has_scales", C.c_int32), ("has_shifts", C.c_int32), ("tps", TpDesc * AA_MAX_LAYERS),
                ("embed_kind", C.c_int32), ("spline_span", C.c_int32), ("env_shared_weights", C.c_int32),
                ("act_kind", C.c_int32 * 3), ("act_consts", C.c_double * 3), ("bessel_convention", C.c_int32)]


class PlanOptions(C.Structure):
    """`aa_plan_options`: kernel-selection switches for A/B measurements and tests (all zero = defaults)."""
    _fields_ = [(n, C.c_int32) for n in (
        "tp_generic", "tp_no_chain", "tp_no_moments", "tp_no_operator", "tp_force_operator", "tp_operator_fused",
        "gemm_no_chain", "gemm_fp32_mfma", "gemm_valu", "gemm_v1", "gemm_lds_epilogue", "f64_column_loop",
        "embed_no_fuse", "fused_forward", "fused_recompute_w0", "moments_waves_per_block", "f64_rows", "no_channel_padding", "fused_tail", "fused_keep_split", "poison_workspace", "no_slot_form", "op_proj_gemm", "op_env_vector", "staged_no_fold", "op_recompute_bvecs", "readout_two_pass", "tp_prefer_moments", "fused_narrow", "chain_staged_weights")]


def options_from_env() -> PlanOptions:
    """The library itself never reads the environment; this host-side mapping keeps the AA_* variables of the
    measurement scripts and tests working (read when a model's plan is created)."""
    env = os.environ
    flag = lambda k: int(env.get(k, "0")[:1] == "1")  # noqa: E731
    o = PlanOptions()
    o.tp_generic, o.tp_no_chain, o.tp_no_moments = flag("AA_TP_GENERIC"), flag("AA_TP_NOCHAIN"), flag("AA_TP_NOMOM")
    o.tp_no_operator, o.tp_force_operator, o.tp_operator_fused = flag("AA_TP_NOOP"), flag("AA_TP_OP"), flag("AA_OP_NOSPLIT")
    o.gemm_no_chain, o.gemm_fp32_mfma, o.gemm_valu, o.gemm_v1 = (flag("AA_GEMM_NOCHAIN"), flag("AA_GEMM_FP32_MFMA"),
                                                                 flag("AA_GEMM_VALU"), flag("AA_GEMM_V1"))
    o.gemm_lds_epilogue = int(env.get("AA_GEMM_DIRECT_EPILOGUE", "1")[:1] == "0")
    o.f64_column_loop = {"0": 1, "2": 2}.get(env.get("AA_F64_NLOOP", "1")[:1], 0)
    o.embed_no_fuse = flag("AA_EMBED_NOFUSE")
    o.fused_forward = {"0": 3, "1": 1, "2": 2, "4": 4}.get(env.get("AA_FUSED", "")[:1], 0)  # unset: automatic; 2 / 4: pure team / mixed form for every graph with segments <= 128 (A/B)
    o.fused_recompute_w0 = flag("AA_FUSED_RECOMPUTE")
    o.moments_waves_per_block = int(env.get("AA_MOM_WPB", "0") or 0)
    o.no_channel_padding = flag("AA_NO_PAD")
    o.fused_tail = {"1": 1, "2": 2}.get(env.get("AA_FUSED_TAIL", "")[:1], 0)  # experimental builds only (AA_BUILD_EXPERIMENTAL=1)
    o.fused_keep_split = int(os.environ.get("AA_FUSED_KEEP", "0") or 0)  # A/B: 1 none, 2 two-body, 3 two-body + lat0 (0: default)
    o.no_slot_form = flag("AA_NO_SLOT_FORM")  # A/B: the unfolded single-layer pipeline on operator-kernel plans
    o.op_proj_gemm = {"1": 1, "0": 2}.get(env.get("AA_OP_PROJ", "")[:1], 0)  # env projections of the operator kernels as batched GEMMs: 1 always, 0 never
    o.op_env_vector = flag("AA_OP_ENV_VECTOR")  # A/B: vector form of tp_op_edge_env in fp64
    o.staged_no_fold = flag("AA_STAGED_NOFOLD")  # A/B: unfolded forward chains in the staged fp32 pipeline
    o.op_recompute_bvecs = flag("AA_OP_RECOMPUTE_BVECS")  # A/B: tp_op_bvecs_kernel in the layer-0 reverse
    o.readout_two_pass = flag("AA_READOUT_TWO_PASS")  # A/B: readout_backward_kernel instead of the fused energy + slope pass
    o.tp_prefer_moments = flag("AA_TP_PREFER_MOM")  # A/B: the round-4 selection (moments kernels) where the operator kernels are now preferred
    o.fused_narrow = {"1": 1, "2": 2, "3": 3, "5": 5, "6": 6, "7": 7}.get(env.get("AA_FUSED_NARROW", "")[:1], 0)  # A/B: 1 = the one-wave-per-SIMD fused forward, 2 = the eight-wave lock-step form, 3 = the four-wave form on small boxes too, 5 = ... with the env projections on the matrix cores
    o.chain_staged_weights = flag("AA_CHAIN_STAGED")  # A/B: the general chain kernel for the one-layer reverse chains
    o.poison_workspace = flag("AA_POISON")  # debugging: NaN-filled workspace before every step
    o.f64_rows = {"0": 2, "2": 1}.get(env.get("AA_F64_ROWS", "")[:1], 0)  # 0: off, 2: wherever applicable
    return o


class RawWeights(C.Structure):
    _fields_ = [("rmax_recip", _dp), ("bessel_weights", _dp), ("center_embed", _dp), ("neighbor_embed", _dp),
                ("basis_linear", _dp), ("embed_mlp", _dp * AA_MAX_MLP_LAYERS), ("env_embed_linear", _dp),
                ("first_proj", _dp), ("latent", (_dp * AA_MAX_MLP_LAYERS) * AA_MAX_LAYERS),
                ("tp_weights", _dp * AA_MAX_LAYERS), ("readout", _dp * AA_MAX_MLP_LAYERS), ("scales", _dp),
                ("shifts", _dp), ("spline_weights", _dp)]


class Graph(C.Structure):
    _fields_ = [("num_atoms", C.c_int64), ("num_edges", C.c_int64), ("center", C.c_void_p), ("nbr", C.c_void_p),
                ("rowptr", C.c_void_p), ("types", C.c_void_p), ("shift_vec", C.c_void_p),
                ("t_rowptr", C.c_void_p), ("t_perm", C.c_void_p), ("atom_begin", C.c_int64), ("atom_end", C.c_int64),
                ("max_degree", C.c_int64)]


class AllegroError(RuntimeError):
    pass


def make_tp_desc(mul, d1, d2, dout, num_paths, coupling, nz_i, nz_j, nz_k, nz_path, nz_val):
    """Returns (TpDesc, keepalive list of numpy arrays)."""
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (nz_i, nz_j, nz_k, nz_path)]
    val = np.ascontiguousarray(nz_val, dtype=np.float64)
    d = TpDesc(int(mul), int(d1), int(d2), int(dout), int(num_paths), int(bool(coupling)), int(len(val)),
               arrs[0].ctypes.data_as(_ip), arrs[1].ctypes.data_as(_ip), arrs[2].ctypes.data_as(_ip),
               arrs[3].ctypes.data_as(_ip), val.ctypes.data_as(_dp))
    return d, arrs + [val]


class NlInput(C.Structure):
    _fields_ = [("num_atoms", C.c_int64), ("pos", C.c_void_p), ("cell", C.c_double * 9), ("pbc", C.c_int32 * 3),
                ("dtype", C.c_int32), ("r_cut", C.c_double)]


class AllegroLib:
    def __init__(self, cdll: C.CDLL, is_emulation: bool = False):
        self.lib = cdll
        self.is_emulation = is_emulation
        L = cdll
        L.aa_last_error.restype = C.c_char_p
        L.aa_version.restype = C.c_int
        L.aa_tp_plan_create.argtypes = [C.POINTER(TpDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.aa_tp_plan_destroy.argtypes = [C.c_void_p]
        L.aa_tp_plan_destroy.restype = None
        L.aa_tp_plan_use_general_kernels.argtypes = [C.c_void_p, C.c_int]
        L.aa_tp_plan_use_general_kernels.restype = C.c_int
        L.aa_tp_plan_is_specialised.argtypes = [C.c_void_p]
        L.aa_tp_plan_is_specialised.restype = C.c_int
        L.aa_tp_forward.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_tp_backward.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_tp_segment_sum.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                        C.c_void_p, C.c_void_p]
        L.aa_tp_weights_workspace_bytes.argtypes = [C.c_void_p, C.c_int64]
        L.aa_tp_weights_workspace_bytes.restype = C.c_size_t
        L.aa_tp_backward_weights.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.aa_linear_wgrad_workspace_bytes.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int]
        L.aa_linear_wgrad_workspace_bytes.restype = C.c_size_t
        L.aa_linear_wgrad.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p]
        L.aa_linear_wgrad.restype = C.c_int
        L.aa_linear_forward_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.aa_linear_forward_workspace_bytes.restype = C.c_size_t
        L.aa_linear_forward.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_int64, C.c_void_p]
        L.aa_linear_forward.restype = C.c_int
        L.aa_weighted_channels.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p]
        L.aa_weighted_channels.restype = C.c_int
        L.aa_weighted_channels_pair.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_weighted_channels_pair.restype = C.c_int
        L.aa_weighted_channels_sum.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.aa_weighted_channels_sum.restype = C.c_int
        L.aa_concat_columns.argtypes = [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_void_p,
                                        C.c_int64, C.c_void_p]
        L.aa_concat_columns.restype = C.c_int
        L.aa_scalar_column.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_scalar_column.restype = C.c_int
        L.aa_silu_derivative.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_silu_derivative.restype = C.c_int
        L.aa_silu_derivative_pair.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_silu_derivative_pair.restype = C.c_int
        L.aa_act_derivative.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_act_derivative.restype = C.c_int
        L.aa_act_derivative_pair.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_act_derivative_pair.restype = C.c_int
        L.aa_debug_gemm_f32.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_debug_gemm_f32.restype = C.c_int
        L.aa_model_plan_create.argtypes = [C.POINTER(ModelConfig), C.POINTER(C.c_void_p)]
        L.aa_model_plan_create_with_options.argtypes = [C.POINTER(ModelConfig), C.POINTER(PlanOptions), C.POINTER(C.c_void_p)]
        L.aa_model_plan_destroy.argtypes = [C.c_void_p]
        L.aa_model_plan_destroy.restype = None
        L.aa_model_plan_set_forward_events.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_model_plan_set_forward_events.restype = C.c_int
        L.aa_model_plan_enable_graph.argtypes = [C.c_void_p, C.c_int]
        L.aa_model_plan_enable_graph.restype = C.c_int
        L.aa_model_plan_enable_taps.argtypes = [C.c_void_p, C.c_int]
        L.aa_model_plan_enable_taps.restype = C.c_int
        L.aa_graph_transpose_workspace_bytes.argtypes = [C.c_int64]
        L.aa_graph_transpose_workspace_bytes.restype = C.c_size_t
        L.aa_graph_transpose.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]
        L.aa_graph_transpose.restype = C.c_int
        L.aa_graph_fingerprint.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.aa_graph_fingerprint.restype = C.c_int
        L.aa_model_file_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.aa_model_file_open.restype = C.c_int
        L.aa_model_file_from_words.argtypes = [C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_void_p)]
        L.aa_model_file_from_words.restype = C.c_int
        L.aa_model_file_config.argtypes = [C.c_void_p]
        L.aa_model_file_config.restype = C.POINTER(ModelConfig)
        L.aa_model_file_weights.argtypes = [C.c_void_p]
        L.aa_model_file_weights.restype = C.POINTER(RawWeights)
        L.aa_model_file_layout_digest.argtypes = [C.c_void_p]
        L.aa_model_file_layout_digest.restype = C.c_uint64
        L.aa_model_file_close.argtypes = [C.c_void_p]
        L.aa_model_file_close.restype = None
        L.aa_model_plan_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.aa_model_plan_describe.restype = C.c_int
        L.aa_model_check.argtypes = [C.c_void_p, C.c_void_p]
        L.aa_model_check.restype = C.c_int
        L.aa_model_virial.argtypes = [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.aa_model_virial.restype = C.c_int
        L.aa_nl_workspace_bytes.argtypes = [C.c_int64]
        L.aa_nl_workspace_bytes.restype = C.c_size_t
        L.aa_nl_count.argtypes = [C.POINTER(NlInput), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
        L.aa_nl_count.restype = C.c_int
        L.aa_nl_fill.argtypes = [C.POINTER(NlInput), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_nl_fill.restype = C.c_int
        L.aa_model_weights_bytes.argtypes = [C.c_void_p]
        L.aa_model_weights_bytes.restype = C.c_size_t
        L.aa_model_plan_layout_hash.argtypes = [C.c_void_p]
        L.aa_model_plan_layout_hash.restype = C.c_uint64
        L.aa_model_pack_weights.argtypes = [C.c_void_p, C.POINTER(RawWeights), C.c_void_p, C.c_size_t, C.c_void_p]
        L.aa_model_workspace_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        L.aa_model_workspace_bytes.restype = C.c_size_t
        L.aa_model_energy_forces.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.aa_model_debug_tap.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]

    def check(self, rc: int, what: str):
        if rc != 0:
            raise AllegroError(f"{what} failed ({rc}): {self.lib.aa_last_error().decode()}")

    # -- tensor-product operator
    def tp_plan_create(self, desc: TpDesc, dtype: int) -> int:
        h = C.c_void_p()
        self.check(self.lib.aa_tp_plan_create(C.byref(desc), dtype, C.byref(h)), "aa_tp_plan_create")
        if os.environ.get("AA_TP_GENERIC", "0")[:1] == "1":  # A/B and tests: the table-driven kernels also for generated signatures
            self.check(self.lib.aa_tp_plan_use_general_kernels(h, 1), "aa_tp_plan_use_general_kernels")
        return h.value

    def tp_plan_destroy(self, h):
        if h:
            self.lib.aa_tp_plan_destroy(h)

    def tp_forward(self, h, E, N, x1, x2, w, rowptr, eids, sf, x2s, out, stream):
        self.check(self.lib.aa_tp_forward(h, E, N, x1, x2, w, rowptr, eids, sf, x2s, out, stream), "aa_tp_forward")

    def tp_backward(self, h, E, N, x1, x2s, w, rowptr, eids, sf, gout, gx1, gx2, stream):
        self.check(self.lib.aa_tp_backward(h, E, N, x1, x2s, w, rowptr, eids, sf, gout, gx1, gx2, stream),
                   "aa_tp_backward")

    def tp_segment_sum(self, dtype_code, E, N, row, x, rowptr, eids, scale, out, stream):
        self.check(self.lib.aa_tp_segment_sum(dtype_code, E, N, row, x, rowptr, eids, scale, out, stream), "aa_tp_segment_sum")

    def tp_backward_weights(self, h, E, N, x1, x2s, rowptr, eids, gout, ws, ws_bytes, gw, stream):
        self.check(self.lib.aa_tp_backward_weights(h, E, N, x1, x2s, rowptr, eids, gout, ws, ws_bytes, gw, stream),
                   "aa_tp_backward_weights")

    # -- model
    def model_plan_create(self, cfg: ModelConfig, options: Optional[PlanOptions] = None) -> int:
        h = C.c_void_p()
        opt = options if options is not None else options_from_env()
        self.check(self.lib.aa_model_plan_create_with_options(C.byref(cfg), C.byref(opt), C.byref(h)), "aa_model_plan_create")
        return h.value

    def model_plan_destroy(self, h):
        if h:
            self.lib.aa_model_plan_destroy(h)


_LIB: Optional[AllegroLib] = None
LIB_NAME = "liballegro_amd.so"


def lib_path() -> str:
    """In-tree library next to this file; ALLEGRO_AMD_LIBRARY points at another build of the same sources
    (instrumented variants of tools/)."""
    from .build import LIB_PATH  # (liballegro_amd_experimental.so under AA_BUILD_EXPERIMENTAL=1)

    return os.environ.get("ALLEGRO_AMD_LIBRARY") or LIB_PATH


def load(build_if_stale: bool = True) -> AllegroLib:
    """Load the gfx950 library; build it in-tree when stale and hipcc exists.  Raises if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if build_if_stale and not os.environ.get("ALLEGRO_AMD_LIBRARY") and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        from .build import build_library

        build_library(verbose=False)
    path = lib_path()
    if not os.path.exists(path):
        raise AllegroError(f"{path} is missing: build it with `python -m allegro_amd.build` (needs hipcc); "
                           "there is no CPU fallback")
    _LIB = AllegroLib(C.CDLL(path))
    return _LIB
