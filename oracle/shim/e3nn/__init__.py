"""ORACLE SHIM (test infrastructure, not product code).

Stand-in for the third-party `e3nn` package, which is a dependency of the
reference (pyproject.toml:15-17 via nequip) but is absent from this container
and from /root/reference.  Only the leaf names the reference imports are
provided.  Semantics are restated from the published e3nn
algorithms (real-basis Wigner 3j from su(2) Clebsch-Gordan + real<->complex
change of basis; component-normalised real spherical harmonics with y as the
polar axis) -- PARITY UNPINNED against e3nn itself: no e3nn golden vectors
exist in /root/reference; the leaves are pinned by identities in
tests/test_conventions.py.
"""
from . import o3  # noqa: F401
