"""ORACLE (test infrastructure, never imported by the product package `allegro_amd`).

Makes the reference's OWN modules importable verbatim from /root/reference behind the
leaf shim in oracle/shim.  Works only in the build container, where
/root/reference is mounted; on the GPU box use oracle/restatement.py + tests/golden.
"""
import os
import sys

REFERENCE_ROOT = "/root/reference"
SHIM_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "allegro"))


def install_shim():
    sys.dont_write_bytecode = True  # never litter /root/reference with __pycache__
    if SHIM_ROOT not in sys.path:
        sys.path.insert(0, SHIM_ROOT)


def import_reference():
    """Returns the reference `allegro` package, imported verbatim from /root/reference."""
    if not reference_available():
        raise RuntimeError("/root/reference is not mounted here (GPU box?) -- use tests/golden fixtures")
    install_shim()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import allegro  # noqa: F401
    import allegro.model  # noqa: F401
    import allegro.nn  # noqa: F401

    return allegro
