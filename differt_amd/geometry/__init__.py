"""Host-side mirror of ``differt.geometry`` for the MI355X hot path (reference names kept)."""

from ._graph import CompleteGraph, DiGraph
from ._mesh import Mesh
from ._paths import LaunchedPaths, TracedPaths, merge_cell_ids
from ._scene import Scene
from ._solver_image_method import (
    consecutive_vertices_are_on_same_side_of_mirror,
    image_method,
    image_of_vertex_with_respect_to_mirror,
    intersection_of_ray_with_plane,
)
from ._solvers import (
    AbstractPathLauncher,
    AbstractPathTracer,
    ExhaustivePathTracer,
    HybridPathTracer,
    SBRPathLauncher,
)
from ._utils import (
    SizedIterator,
    assemble_path,
    cartesian_to_spherical,
    fibonacci_lattice,
    first_triangle_hit_by_ray,
    generate_all_path_candidates,
    generate_all_path_candidates_chunks_iter,
    generate_all_path_candidates_iter,
    normalize,
    ray_intersect_any_triangle,
    ray_intersect_triangle,
    spherical_to_cartesian,
    triangles_visible_from_vertex,
    viewing_frustum,
)

__all__ = [
    "AbstractPathLauncher",
    "AbstractPathTracer",
    "CompleteGraph",
    "DiGraph",
    "ExhaustivePathTracer",
    "HybridPathTracer",
    "LaunchedPaths",
    "Mesh",
    "SBRPathLauncher",
    "Scene",
    "SizedIterator",
    "TracedPaths",
    "assemble_path",
    "cartesian_to_spherical",
    "fibonacci_lattice",
    "spherical_to_cartesian",
    "triangles_visible_from_vertex",
    "viewing_frustum",
    "consecutive_vertices_are_on_same_side_of_mirror",
    "first_triangle_hit_by_ray",
    "generate_all_path_candidates",
    "generate_all_path_candidates_chunks_iter",
    "generate_all_path_candidates_iter",
    "image_method",
    "merge_cell_ids",
    "image_of_vertex_with_respect_to_mirror",
    "intersection_of_ray_with_plane",
    "normalize",
    "ray_intersect_any_triangle",
    "ray_intersect_triangle",
]
