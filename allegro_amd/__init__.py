"""allegro_amd -- MI355X (gfx950) native implementation of the Allegro hot path.

Public surface (host-side mirrors of the reference's operator / module interface; kernels are hand-written HIP
behind the C ABI of include/allegro_amd.h):

    HipContracter, enable_HipContracter      drop-in for allegro.nn._strided.Contracter and its model modifier
    HipAllegroModel                          energies / forces / virial of allegro.model.AllegroModel
    neighbor_list, PreparedGraph             on-device cell list -> center-sorted CSR graph
    build_library                            (re)build allegro_amd/liballegro_amd.so with hipcc

Importing the package never touches the GPU or the compiler; the shared library is loaded (and, when stale, built
in-tree) on first use, and every op raises if it is missing -- there is no CPU fallback.
"""
__version__ = "0.2.0"

from .build import build_library  # noqa: E402,F401
from .nn import (HipAllegroModel, HipContracter, PreparedGraph, enable_HipContracter, neighbor_list)  # noqa: E402,F401

# nequip plugin hook-up (no-op when nequip is not installed); also the target of the `nequip.extension` entry point
from ._nequip_ext import register as _register_nequip_extension  # noqa: E402

_register_nequip_extension()

__all__ = ["HipContracter", "enable_HipContracter", "HipAllegroModel", "PreparedGraph", "neighbor_list",
           "build_library", "__version__"]
