"""CPU: the gfx950 shared library builds/loads and exports every symbol include/allegro_amd.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

from allegro_amd import _lib
from allegro_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "allegro_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aa_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    path = build_library(verbose=False)
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/allegro_amd.h but not exported"
    assert lib.aa_version() >= 1


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden + the header's visibility pragma + the linker version script csrc/exports.map: no mangled `aa::`
    helper, kernel handle or libstdc++ template instantiation leaks out of the shared object -- the
    dynamic symbol table's defined symbols are exactly the header's functions."""
    import subprocess

    path = build_library(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    defined = sorted(l.split()[-1] for l in out.splitlines() if l.split())
    extra = [s for s in defined if s not in set(declared_symbols())]
    assert not extra, f"exported beyond include/allegro_amd.h: {extra[:8]}"


def test_package_ships_the_public_header():
    # pyproject package-data: an installed copy rebuilds itself from allegro_amd/csrc + allegro_amd/include alone
    from allegro_amd.build import INCLUDE_DIR

    pub = open(os.path.join(ROOT, "include", "allegro_amd.h"), "rb").read()
    assert open(os.path.join(INCLUDE_DIR, "allegro_amd.h"), "rb").read() == pub
    assert '"include/*.h"' in open(os.path.join(ROOT, "pyproject.toml")).read()


def test_ctypes_structs_match_header_sizes():
    # layout sanity of the ctypes mirrors (pointer-heavy structs: check field counts against the header)
    assert ctypes.sizeof(_lib.TpDesc) == 7 * 4 + 4 + 5 * 8  # 7 int32 + pad + 5 pointers
    assert ctypes.sizeof(_lib.Graph) == 12 * 8  # 2 int64 + 7 pointers (incl. the optional transposed CSR) + owned range + max_degree
    assert _lib.ModelConfig.tps.size == _lib.AA_MAX_LAYERS * ctypes.sizeof(_lib.TpDesc)


def test_product_refuses_cpu_tensors():
    import pytest
    import torch

    from allegro_amd.nn import HipContracter

    c = HipContracter("0e+1o", "0e+1o", "0e+1o", mul=4)
    x = torch.randn(5, 4, 4)
    with pytest.raises(Exception):
        c(x, x, torch.zeros(5, dtype=torch.long), 1)
