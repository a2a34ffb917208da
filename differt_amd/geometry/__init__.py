"""Host-side mirror of ``differt.geometry`` for the MI355X hot path (reference names kept)."""

from ._utils import (
    SizedIterator,
    assemble_path,
    first_triangle_hit_by_ray,
    generate_all_path_candidates,
    generate_all_path_candidates_chunks_iter,
    generate_all_path_candidates_iter,
    normalize,
    ray_intersect_any_triangle,
    ray_intersect_triangle,
)

__all__ = [
    "SizedIterator",
    "assemble_path",
    "first_triangle_hit_by_ray",
    "generate_all_path_candidates",
    "generate_all_path_candidates_chunks_iter",
    "generate_all_path_candidates_iter",
    "normalize",
    "ray_intersect_any_triangle",
    "ray_intersect_triangle",
]
