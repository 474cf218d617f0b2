"""Shared helpers for the parity tests of the HIP path."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from allegro_amd import _lib  # noqa: E402
from allegro_amd.nn import HipAllegroModel  # noqa: E402

_EMU = None


def emu_lib() -> _lib.AllegroLib:
    """TEST-ONLY CPU emulation build of the same HIP sources (tests/emu)."""
    global _EMU
    if _EMU is None:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from build_emu import build_emu

        _EMU = _lib.AllegroLib(ctypes.CDLL(build_emu()), is_emulation=True)
    return _EMU


def model_from_fixture(fx, dtype, lib=None, device="cpu") -> HipAllegroModel:
    cfg = dict(fx["cfg"])
    cfg["model_dtype"] = {torch.float32: "float32", torch.float64: "float64"}[dtype]
    m = HipAllegroModel(**cfg)
    m.load_state_dict({"func." + k: v.to(dtype) if v.is_floating_point() else v for k, v in fx["sd"].items()})
    m = m.to(device)
    if lib is not None:
        m._bind_library(lib)
    return m


def fixture_data(fx, dtype, device="cpu"):
    d = {"pos": fx["pos"].to(dtype).to(device), "edge_index": fx["edge_index"].to(device),
         "atom_types": fx["types"].to(device)}
    return d, (None if fx["shift_vec"] is None else fx["shift_vec"].to(dtype).to(device))
