"""EM post-processing of traced paths on the MI355X (second half).

Mirrors the part of ``differt.em`` that ``differt.plugins.deepmimo.export`` composes:
constants (em/_constants.py), ``length_to_delay`` / ``path_delay`` / ``sp_directions`` /
``sp_rotation_matrix`` / ``fspl`` (em/_utils.py), ``refractive_index`` / ``fresnel_coefficients`` /
``reflection_coefficients`` / ``refraction_coefficients`` (em/_fresnel.py) and a minimal ``Material``
(em/_material.py).  Antennas, UTD and the full ITU table are out of scope.
"""

from ._constants import c, epsilon_0, mu_0, z_0
from ._fresnel import (
    fresnel_coefficients,
    reflection_coefficients,
    refraction_coefficients,
    refractive_index,
)
from ._material import Material, materials
from ._utils import fspl, length_to_delay, path_delay, sp_directions, sp_rotation_matrix

__all__ = [
    "Material",
    "c",
    "epsilon_0",
    "fresnel_coefficients",
    "fspl",
    "length_to_delay",
    "materials",
    "mu_0",
    "path_delay",
    "reflection_coefficients",
    "refraction_coefficients",
    "refractive_index",
    "sp_directions",
    "sp_rotation_matrix",
    "z_0",
]
