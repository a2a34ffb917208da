"""Deprecated alias package, like ``differt.rt`` in the reference
(``differt/src/differt/rt/__init__.py:1-45``), including the pre-0.10 plural operator names
(``CHANGELOG.md:46-52``) that BASELINE.json's north-star still uses."""

from ..geometry import *  # noqa: F403
from ..geometry import (
    first_triangle_hit_by_ray as first_triangles_hit_by_rays,
    ray_intersect_any_triangle as rays_intersect_any_triangle,
    ray_intersect_triangle as rays_intersect_triangles,
)
from ..geometry._graph import CompleteGraph, DiGraph  # noqa: F401  (differt_core.rt names)
