"""A minimal ``Material`` (name, electrical properties as a function of frequency, thickness) with
the interface ``differt.plugins.deepmimo.export`` relies on (em/_material.py:19-82): the material
DATABASE of the reference (the ITU-R P.2040 table, em/_material.py:340-420) is data and is not
reproduced; ``from_itu_properties`` builds an entry from the recommendation's (a, b, c, d) model
``eta_r = a f_GHz^b``, ``sigma = c f_GHz^d`` (em/_material.py:134-156)."""

from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass

__all__ = ["Material", "materials"]


@dataclass(frozen=True)
class Material:
    name: str
    properties: Callable[[float], tuple[float, float]]
    thickness: float | None = None
    aliases: tuple[str, ...] = ()

    def relative_permittivity(self, frequency: float) -> float:
        return self.properties(frequency)[0]

    def conductivity(self, frequency: float) -> float:
        return self.properties(frequency)[1]

    @classmethod
    def from_itu_properties(cls, name: str, a: float, b: float, c: float, d: float,
                            f_range_ghz: tuple[float, float] | None = None,
                            thickness: float | None = None) -> "Material":
        def properties(frequency: float) -> tuple[float, float]:
            f_ghz = float(frequency) * 1e-9
            if f_range_ghz is not None and not (f_range_ghz[0] <= f_ghz <= f_range_ghz[1]):
                return -1.0, -1.0  # em/_material.py:152-154: outside the validity range
            return a * f_ghz**b, c * f_ghz**d

        return cls(name, properties, thickness, (f"itu_{name.lower().replace(' ', '_')}",))


class _Materials(dict):
    """Name or ``itu_*`` alias -> material (em/_material.py:233-325)."""

    def __missing__(self, key):
        for m in self.values():
            if key in m.aliases:
                return m
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or any(key in m.aliases for m in self.values())


# Three entries of Recommendation ITU-R P.2040 (first frequency range each), enough for the mirrored
# tests; extend with Material.from_itu_properties.
materials = _Materials({
    "Vacuum": Material.from_itu_properties("Vacuum", 1.0, 0.0, 0.0, 0.0),
    "Concrete": Material.from_itu_properties("Concrete", 5.24, 0.0, 0.0462, 0.7822, (1.0, 100.0)),
    "Glass": Material.from_itu_properties("Glass", 6.27, 0.0, 0.0043, 1.1925, (0.1, 100.0)),
})
