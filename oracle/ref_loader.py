"""ORACLE (test infrastructure, never imported by the product package `allegro_amd`).

Makes the reference's OWN modules importable verbatim from /root/reference.  Its third-party leaves (e3nn, nequip, hydra)
are absent from this image, so a leaf shim (oracle/shim) stands in for them -- but ONLY for the packages
that are really missing: where the real `e3nn` / `nequip` are installed they win, the shim fills the holes
(`AA_ORACLE_FORCE_SHIM=1` forces the shim, e.g. to regenerate the committed fixtures bit-identically).
tests/test_real_leaves.py compares shim and real leaves whenever both exist.  Works only where /root/reference is mounted
(or the reference is pip-installed as `allegro`); on the GPU box use oracle/restatement.py + tests/golden.
"""
import importlib.machinery
import os
import sys

REFERENCE_ROOT = "/root/reference"
SHIM_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")
LEAVES = ("e3nn", "nequip", "hydra")


def real_package(name: str):
    """Path of an installed top-level package `name` that is NOT the shim (searched on sys.path without the shim root), or None."""
    path = [p for p in sys.path if os.path.abspath(p or ".") != SHIM_ROOT]
    spec = importlib.machinery.PathFinder.find_spec(name, path)
    if spec is None or spec.origin is None:
        return None
    return None if os.path.abspath(spec.origin).startswith(SHIM_ROOT + os.sep) else spec.origin


def real_leaves() -> dict:
    return {n: real_package(n) for n in LEAVES}


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "allegro")) or real_package("allegro") is not None


def install_shim():
    """Puts oracle/shim on sys.path: in FRONT when no real leaf exists (or AA_ORACLE_FORCE_SHIM=1), otherwise at the END, so that
    installed packages are preferred and the shim only supplies what is missing."""
    sys.dont_write_bytecode = True  # never litter /root/reference with __pycache__
    force = os.environ.get("AA_ORACLE_FORCE_SHIM", "0")[:1] == "1"
    have_real = any(v for k, v in real_leaves().items() if k != "hydra")
    if SHIM_ROOT in sys.path:
        sys.path.remove(SHIM_ROOT)
    if force or not have_real:
        sys.path.insert(0, SHIM_ROOT)
    else:
        sys.path.append(SHIM_ROOT)


def import_reference():
    """Returns the reference `allegro` package, imported verbatim from /root/reference (or the installed one)."""
    if not reference_available():
        raise RuntimeError("/root/reference is not mounted here (GPU box?) -- use tests/golden fixtures")
    install_shim()
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "allegro")) and REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import allegro  # noqa: F401
    import allegro.model  # noqa: F401
    import allegro.nn  # noqa: F401

    return allegro


def load_shim_package(name: str):
    """The SHIM's copy of leaf package `name` under the alias `aa_shim_<name>` (its modules use relative imports), importable
    next to the real package of the same name: what tests/test_real_leaves.py compares."""
    import importlib.util

    alias = "aa_shim_" + name
    if alias in sys.modules:
        return sys.modules[alias]
    root = os.path.join(SHIM_ROOT, name)
    spec = importlib.util.spec_from_file_location(alias, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod
