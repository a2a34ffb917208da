"""ctypes binding of ``libdiffert_amd.so`` (the C ABI declared in ``include/differt_amd.h``).

The product path has NO fallback: if the HIP library is missing or a call fails, we raise.
"""

from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("DIFFERT_AMD_LIB", _HERE / "lib" / "libdiffert_amd.so"))
HEADER_PATH = _HERE.parent / "include" / "differt_amd.h"

DRT_OK = 0
DRT_E_INVALID, DRT_E_HIP, DRT_E_NO_DEVICE, DRT_E_CAPACITY, DRT_E_OVERFLOW, DRT_E_UNSUPPORTED = (
    -1,
    -2,
    -3,
    -4,
    -5,
    -6,
)
DRT_MAX_ORDER = 8


class DrtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdiffert_amd error {code}: {msg}")
        self.code = code
        self.msg = msg


class CapacityError(DrtError):
    """A caller-provided capacity (survivor queue / output rows) was too small."""


class TraceStats(C.Structure):
    """``drt_trace_stats``: per-stage counters + HIP-event timers of one compact trace."""

    _fields_ = [
        ("candidates", C.c_int64),
        ("survivors", C.c_int64),
        ("valid", C.c_int64),
        ("filter_ms", C.c_float),
        ("occlusion_ms", C.c_float),
        ("sort_emit_ms", C.c_float),
        ("reserved", C.c_int32),
    ]


class TraceParams(C.Structure):
    _fields_ = [
        ("epsilon", C.c_float),
        ("hit_tol", C.c_float),
        ("min_len", C.c_float),
        ("flags", C.c_int32),
        ("stats", C.POINTER(TraceStats)),
    ]


DRT_TRACE_USE_BVH = 1
DRT_TRACE_SKIP_OCCLUSION = 2
DRT_TRACE_DETERMINISTIC_GRAD = 4
DRT_TRACE_OVERFLOW_SURVIVORS, DRT_TRACE_OVERFLOW_PATHS = 1, 2
ABI_VERSION = 7  # DRT_ABI_VERSION of include/differt_amd.h this binding was written against


class BeamStats(C.Structure):
    """``drt_beam_stats``."""

    _fields_ = [
        ("levels", C.c_int64 * 4),
        ("rows", C.c_int64),
        ("slices", C.c_int64),
        ("valid", C.c_int64),
        ("grazing_prefixes", C.c_int64),
        ("unit_m", C.c_float),
        ("magnitude", C.c_float),
        ("pair_mode", C.c_int32),
        ("paired_primitives", C.c_int32),
        ("expand_last_ms", C.c_float),
        ("emit_ms", C.c_float),
        ("trace_ms", C.c_float),
        ("next_probe_prefixes", C.c_float),
    ]


class BeamParams(C.Structure):
    """``drt_beam_params``."""

    _fields_ = [
        ("kappa", C.c_float),
        ("flags", C.c_int32),
        ("max_entries", C.c_int64),
        ("max_records", C.c_int64),
        ("max_rows", C.c_int64),
        ("max_survivors", C.c_int64),
        ("probe_prefixes", C.c_int64),
        ("shard_rank", C.c_int64),
        ("shard_world", C.c_int64),
        ("stats", C.POINTER(BeamStats)),
    ]


DRT_BEAM_EXPAND_PLAIN, DRT_BEAM_EMIT_PLAIN, DRT_BEAM_EMIT_CLUSTERED, DRT_BEAM_NO_PAIRS, DRT_BEAM_ROWS_PLAIN = 1, 2, 4, 8, 16
DRT_BEAM_EXPAND_FUSED = 32
DRT_BEAM_OVERFLOW_ENTRIES, DRT_BEAM_OVERFLOW_RECORDS, DRT_BEAM_OVERFLOW_ROWS = 4, 8, 16
DRT_HYBRID_PREFIX, DRT_HYBRID_RAGGED = 1, 2
DRT_CAND_PACKED_KEYS = 4
DRT_CAND_PAIR_BLOCKS = 8


class EmParams(C.Structure):
    _fields_ = [
        ("frequency", C.c_double),
        ("tx_polarization", C.c_int32),
        ("tx_vector", C.c_float * 3),
        ("rx_polarization", C.c_int32),
        ("rx_vector", C.c_float * 3),
    ]


class Candidates(C.Structure):
    _fields_ = [
        ("table", C.c_void_p),
        ("num_candidates", C.c_int64),
        ("rank_lo", C.c_int64),
        ("num_nodes", C.c_int64),
        ("node_map", C.c_void_p),
        ("order", C.c_int32),
        ("reserved", C.c_int32),
        ("first_map", C.c_void_p),
        ("num_first", C.c_int64),
        ("last_map", C.c_void_p),
        ("num_last", C.c_int64),
        ("pair_offsets", C.c_void_p),
        ("first_offsets", C.c_void_p),
        ("last_offsets", C.c_void_p),
    ]


_vp, _i64, _u64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_float, C.c_size_t

# name -> (restype, argtypes); restype int32 means "status code, checked"
_SIGNATURES = {
    "drt_abi_version": (_i32, []),
    "drt_last_error": (C.c_char_p, []),
    "drt_device_check": (_i32, []),
    "drt_ray_intersect_triangle_dense": (_i32, [_vp, _vp, _i64, _vp, _i64, _f32, _vp, _vp, _vp]),
    "drt_ray_intersect_triangle_paired": (_i32, [_vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "drt_ray_intersect_triangle_dense_batched": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp]),
    "drt_ray_intersect_any_triangle": (
        _i32,
        [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _f32, _f32, _vp, _vp],
    ),
    "drt_first_triangle_hit_by_ray_workspace_size": (_sz, [_i64]),
    "drt_first_triangle_hit_by_ray": (
        _i32,
        [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _f32, _i64, _vp, _vp, _vp, _sz, _vp],
    ),
    "drt_first_hit_keys": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _f32, _i64, _vp, _i32, _vp]),
    "drt_first_hit_finalize": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "drt_first_hit_vjp": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "drt_normalize": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "drt_image_of_vertex": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "drt_intersection_of_ray_with_plane": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "drt_image_method": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "drt_image_method_vjp": (
        _i32,
        [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    ),
    "drt_image_method_strided": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp]),
    "drt_image_method_vjp_strided": (
        _i32,
        [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    ),
    "drt_consecutive_vertices_same_side": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "drt_mesh_build_bvh": (_i32, [_vp, _vp]),
    "drt_mesh_has_bvh": (_i32, [_vp]),
    "drt_mesh_ray_intersect_any_triangle": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "drt_mesh_first_triangle_hit_by_ray": (_i32, [_vp, _vp, _vp, _i64, _f32, _i64, _vp, _vp, _vp]),
    "drt_mesh_triangles_visible_from_vertex": (_i32, [_vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp]),
    "drt_mesh_triangles_visible_samples": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp]),
    "drt_viewing_frustum": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "drt_viewing_frustum_points": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "drt_viewing_frustum_general": (_i32, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "drt_launch_paths": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i32, _f32, _i64, _f32, _vp, _vp, _vp, _vp]),
    "drt_fibonacci_lattice": (_i32, [_i64, _vp, _vp, _vp]),
    "drt_triangles_visible_from_vertex": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _f32, _vp, _vp, _vp]),
    "drt_mesh_create": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, C.POINTER(_vp)]),
    "drt_mesh_destroy": (_i32, [_vp]),
    "drt_mesh_num_triangles": (_i64, [_vp]),
    "drt_mesh_triangle_vertices": (_vp, [_vp]),
    "drt_mesh_normals": (_vp, [_vp]),
    "drt_mesh_copy": (_i32, [_vp, _vp, _vp, _vp]),
    "drt_complete_graph_count": (_i32, [_u64, _u64, _u64, _u64, C.POINTER(_u64), C.POINTER(_i32)]),
    "drt_complete_graph_count_exact": (_i32, [_u64, _u64, _u64, _u64, C.POINTER(_u64), C.POINTER(_i32)]),
    "drt_complete_graph_fill_host": (_i32, [_u64, _u64, _u64, _u64, _i32, _u64, _u64, _vp]),
    "drt_candidates_fill": (_i32, [_i64, _i32, _i64, _i64, _vp, _i32, _vp, _vp]),
    "drt_digraph_from_complete_graph": (_i32, [_u64, C.POINTER(_vp)]),
    "drt_digraph_from_adjacency_matrix": (_i32, [_vp, _u64, C.POINTER(_vp)]),
    "drt_digraph_destroy": (_i32, [_vp]),
    "drt_digraph_num_nodes": (_u64, [_vp]),
    "drt_digraph_insert_from_and_to_nodes": (
        _i32,
        [_vp, _i32, _vp, _vp, C.POINTER(_u64), C.POINTER(_u64)],
    ),
    "drt_digraph_filter_by_mask": (_i32, [_vp, _vp, _u64, _i32]),
    "drt_digraph_disconnect_nodes": (_i32, [_vp, _vp, _u64, _i32]),
    "drt_digraph_iter_create": (_i32, [_vp, _u64, _u64, _u64, _i32, C.POINTER(_vp)]),
    "drt_digraph_iter_destroy": (_i32, [_vp]),
    "drt_digraph_iter_next_chunk": (_i32, [_vp, _u64, _vp, C.POINTER(_u64)]),
    "drt_trace_dense_workspace_size": (_sz, [_i64, _i64, _i64]),
    "drt_trace_paths_dense": (
        _i32,
        [_vp, C.POINTER(TraceParams), _vp, _i64, _vp, _i64, C.POINTER(Candidates), _vp, _vp, _vp,
         _vp, _sz, _vp],
    ),
    "drt_trace_paths_dense_ex": (
        _i32,
        [_vp, C.POINTER(TraceParams), _vp, _i64, _vp, _i64, C.POINTER(Candidates), _vp, _vp, _vp, _vp,
         _vp, _vp, _sz, _vp],
    ),
    "drt_trace_dense_capped_workspace_size": (_sz, [_i64]),
    "drt_trace_paths_dense_capped": (
        _i32,
        [_vp, C.POINTER(TraceParams), _vp, _i64, _vp, _i64, C.POINTER(Candidates), _vp, _vp, _vp, _vp,
         _vp, _i64, _vp, _sz, _vp],
    ),
    "drt_trace_compact_workspace_size": (_sz, [_i64, _i64]),
    "drt_comm_unique_id": (_i32, [_vp]),
    "drt_comm_init": (_i32, [_vp, _i32, _i32, _vp]),
    "drt_comm_destroy": (_i32, [_vp]),
    "drt_comm_rank": (_i32, [_vp]),
    "drt_comm_world": (_i32, [_vp]),
    "drt_allreduce_min_u64": (_i32, [_vp, _vp, _i64, _vp]),
    "drt_allreduce_max_u8": (_i32, [_vp, _vp, _i64, _vp]),
    "drt_allreduce_sum_f32": (_i32, [_vp, _vp, _i64, _vp]),
    "drt_allgather_bytes": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "drt_trace_vjp_workspace_size": (_sz, [_i64, _i32]),
    "drt_trace_paths_vjp_ex": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drt_sort_u64_workspace_size": (_sz, [_i64, _i32]),
    "drt_sort_u64": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _sz, _vp]),
    "drt_mesh_build_beam_clusters": (_i32, [_vp, _vp]),
    "drt_mesh_build_beam_clusters_ex": (_i32, [_vp, _i32, _vp]),
    "drt_mesh_beam_pairing": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "drt_mesh_beam_pairing_table": (_i32, [_vp, _vp, _i64, _vp]),
    "drt_trace_beam_workspace_size": (_sz, [_i64, _i64, _i64, _i32, _vp, _i64]),
    "drt_trace_paths_beam": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drt_trace_hybrid_pairs_workspace_size": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "drt_trace_paths_hybrid_pairs": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp,
                                          C.POINTER(_i64), C.POINTER(_i64), _vp, _sz, _vp]),
    "drt_trace_paths_beam_async": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drt_trace_paths_compact": (
        _i32,
        [_vp, C.POINTER(TraceParams), _vp, _i64, _vp, _i64, C.POINTER(Candidates), _i64, _i64, _vp,
         _vp, _vp, C.POINTER(_i64), _vp, _sz, _vp],
    ),
    "drt_ray_intersect_triangle_vjp": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "drt_launch_paths_vjp": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "drt_warp_ray_prep": (_i32, [_vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp]),
    "drt_warp_first_hit_finish": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "drt_trace_paths_compact_async": (
        _i32,
        [_vp, C.POINTER(TraceParams), _vp, _i64, _vp, _i64, C.POINTER(Candidates), _i64, _i64, _vp,
         _vp, _vp, _vp, _vp, _sz, _vp],
    ),
    "drt_trace_paths_vjp": (
        _i32,
        [_vp, _vp, _i64, _vp, _i64, C.POINTER(Candidates), _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    ),
    "drt_ray_intersect_triangle_smooth": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp]),
    "drt_ray_intersect_triangle_smooth_vjp": (
        _i32, [_vp, _vp, _i64, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "drt_ray_intersect_any_triangle_smooth": (
        _i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _f32, _f32, _f32, _i64, _vp, _vp]),
    "drt_ray_intersect_any_triangle_smooth_vjp": (
        _i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _f32, _f32, _f32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "drt_consecutive_vertices_same_side_smooth": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp]),
    "drt_trace_paths_dense_smooth": (
        _i32, [_vp, C.POINTER(TraceParams), _f32, _i64, _vp, _i64, _vp, _i64, C.POINTER(Candidates),
               _vp, _vp, _vp, _vp]),
    "drt_trace_paths_dense_smooth_vjp": (
        _i32, [_vp, C.POINTER(TraceParams), _f32, _i64, _vp, _i64, _vp, _i64, C.POINTER(Candidates),
               _vp, _vp, _vp, _vp, _vp, _vp]),
    "drt_cartesian_to_spherical": (_i32, [_vp, _i64, _vp, _vp]),
    "drt_spherical_to_cartesian": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "drt_path_length": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "drt_length_to_delay": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "drt_fspl": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "drt_refractive_index": (_i32, [_vp, _i64, _vp, _vp]),
    "drt_sp_directions": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "drt_sp_rotation_matrix": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "drt_fresnel_coefficients": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "drt_complex_refractive_index": (_i32, [_vp, _vp, _i64, C.c_double, _vp]),
    "drt_paths_channel": (
        _i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, C.POINTER(EmParams),
               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "drt_row_cell_ids_workspace_size": (_sz, [_i64]),
    "drt_row_cell_ids": (_i32, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "drt_paths_channel_vjp": (
        _i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, C.POINTER(EmParams), _vp, _vp, _vp]),
}

# functions whose int32 result is NOT a status code
_NOT_STATUS = {"drt_abi_version", "drt_mesh_has_bvh", "drt_comm_rank", "drt_comm_world", "drt_mesh_beam_pairing"}

_LIB = None


def declared_symbols() -> list[str]:
    """Every function name declared in include/differt_amd.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drt_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the shared library (raises if it has not been built: no CPU fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m differt_amd.build` "
            "(hipcc, gfx950).  differt_amd has no CPU fallback."
        )
    L = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is None:  # reported by tests/test_abi.py and __graft_entry__.build(); calling it raises
            continue
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def check(code: int) -> None:
    if code == DRT_OK:
        return
    msg = load().drt_last_error().decode(errors="replace")
    if code == DRT_E_CAPACITY:
        raise CapacityError(code, msg)
    if code == DRT_E_INVALID:
        raise ValueError(msg)
    raise DrtError(code, msg)


def call(name: str, *args):
    """Call a status-returning entry point and raise on error."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if _SIGNATURES[name][0] is _i32 and name not in _NOT_STATUS:
        check(rc)
    return rc


def require_device() -> None:
    """Raise unless a gfx950 GPU is usable (the product never runs on CPU)."""
    check(load().drt_device_check())
