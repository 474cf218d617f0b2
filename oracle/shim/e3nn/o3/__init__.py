from ._irreps import Irrep, Irreps
from ._wigner import wigner_3j
from ._spherical_harmonics import SphericalHarmonics, spherical_harmonics

__all__ = ["Irrep", "Irreps", "wigner_3j", "SphericalHarmonics", "spherical_harmonics"]
