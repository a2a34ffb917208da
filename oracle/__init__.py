"""CPU oracle for the DiffeRT hot path -- TEST INFRASTRUCTURE ONLY.

A NumPy/ctypes front-end over ``differt_oracle.c`` (plain C, float32, no FMA contraction), the
line-by-line restatement of the reference algorithm.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package;
``differt_amd`` (the product) never does.

Parity status: pinned against the reference's own golden vectors (``tests/golden``), see
``tests/test_oracle_golden.py``.  Function names and argument meaning mirror the reference
(`/root/reference/differt/src/differt/geometry/_utils.py`, `_solver_image_method.py`,
`_solvers.py`; `differt-core/src/geometry/graph.rs`).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

F32_EPS = float(np.finfo(np.float32).eps)
DEFAULT_EPSILON = 10.0 * F32_EPS  # UT:1257-1259
DEFAULT_HIT_TOL = 100.0 * F32_EPS  # UT:1418-1420
DEFAULT_MIN_LEN = 10.0 * F32_EPS  # SV:514-516


def build(native: bool = False) -> Path:
    """Compile the oracle with gcc (``make -C oracle``); returns the .so path."""
    target = "native" if native else "all"
    subprocess.run(["make", "-C", str(_HERE), target], check=True, capture_output=True)
    name = "libdiffert_oracle_native.so" if native else "libdiffert_oracle.so"
    return _HERE / "_build" / name


class _TraceParams(C.Structure):
    _fields_ = [
        ("epsilon", C.c_float),
        ("hit_tol", C.c_float),
        ("min_len", C.c_float),
        ("assume_quads", C.c_int32),
    ]


def lib(path: os.PathLike | None = None):
    """Load (building if necessary) the oracle shared library."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    env = os.environ.get("DRT_ORACLE_LIB")  # e.g. the sanitizer build (`make -C oracle sanitize`)
    so = Path(path) if path is not None else (Path(env) if env else _HERE / "_build" / "libdiffert_oracle.so")
    if not so.exists():
        if env and path is None:
            raise FileNotFoundError(f"DRT_ORACLE_LIB={env} does not exist")
        so = build()
    L = C.CDLL(str(so))
    vp, i64, f32, i32 = C.c_void_p, C.c_int64, C.c_float, C.c_int32
    u64 = C.c_uint64
    sig = {
        "orc_normalize": (None, [vp, i64, vp, vp]),
        "orc_triangle_vertices": (None, [vp, vp, i64, vp]),
        "orc_mesh_normals": (None, [vp, i64, vp]),
        "orc_ray_intersect_triangle_paired": (None, [vp, vp, vp, i64, f32, vp, vp]),
        "orc_ray_intersect_triangle_dense": (None, [vp, vp, i64, vp, i64, f32, vp, vp]),
        "orc_ray_intersect_any_triangle": (
            None,
            [vp, vp, i64, vp, i64, i64, vp, i64, f32, f32, vp],
        ),
        "orc_first_triangle_hit_by_ray": (
            None,
            [vp, vp, i64, vp, i64, i64, vp, i64, f32, i64, vp, vp],
        ),
        "orc_image_of_vertex": (None, [vp, vp, vp, i64, vp]),
        "orc_intersection_of_ray_with_plane": (None, [vp, vp, vp, vp, i64, vp]),
        "orc_image_method": (None, [vp, vp, vp, vp, i64, C.c_int, vp]),
        "orc_same_side_of_mirror": (None, [vp, vp, vp, i64, C.c_int, vp]),
        "orc_trace_path_candidates": (
            C.c_int,
            [vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, C.c_int, vp, vp, vp, vp, vp],
        ),
        "orc_cg_iter_new": (vp, [u64, u64, u64, u64, C.c_int]),
        "orc_cg_iter_free": (None, [vp]),
        "orc_cg_iter_len": (u64, [vp]),
        "orc_cg_iter_overflowed": (C.c_int, [vp]),
        "orc_cg_iter_path_depth": (u64, [vp]),
        "orc_cg_iter_next": (C.c_int, [vp, vp]),
        "orc_cg_iter_collect": (u64, [vp, vp, u64]),
        "orc_digraph_from_adjacency_matrix": (vp, [vp, u64]),
        "orc_digraph_from_complete_graph": (vp, [u64]),
        "orc_digraph_free": (None, [vp]),
        "orc_digraph_num_nodes": (u64, [vp]),
        "orc_digraph_insert_from_and_to_nodes": (None, [vp, C.c_int, vp, vp, vp, vp]),
        "orc_digraph_filter_by_mask": (C.c_int, [vp, vp, u64, C.c_int]),
        "orc_digraph_disconnect_nodes": (C.c_int, [vp, vp, u64, C.c_int]),
        "orc_dg_iter_new": (vp, [vp, u64, u64, u64, C.c_int]),
        "orc_dg_iter_free": (None, [vp]),
        "orc_dg_iter_next": (C.c_int, [vp, vp]),
        "orc_dg_iter_path_depth": (u64, [vp]),
        "orc_dg_iter_collect": (u64, [vp, vp, u64]),
        "orc_differentiable_distance": (None, [vp, vp, vp, vp, i64, vp]),
        "orc_abi_version": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = L
    return L


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------
def normalize(vectors):
    """UT:29-72.  Returns (unit vectors, lengths)."""
    v = _f32(vectors)
    out = np.empty_like(v)
    lens = np.empty(v.shape[:-1], dtype=np.float32)
    lib().orc_normalize(_p(v), v.size // 3, _p(out), _p(lens))
    return out, lens


def triangle_vertices(vertices, triangles):
    """ME:899-905."""
    v = _f32(vertices)
    t = np.ascontiguousarray(np.asarray(triangles, dtype=np.int32))
    out = np.empty((t.shape[0], 3, 3), dtype=np.float32)
    if t.shape[0]:
        lib().orc_triangle_vertices(_p(v), _p(t), t.shape[0], _p(out))
    return out


def mesh_normals(tv):
    """ME:950-956."""
    tv = _f32(tv)
    out = np.empty((tv.shape[0], 3), dtype=np.float32)
    if tv.shape[0]:
        lib().orc_mesh_normals(_p(tv), tv.shape[0], _p(out))
    return out


def box_mesh(length=1.0, width=1.0, height=1.0, *, with_top=False, with_bottom=True):
    """ME:2109-2217 ``Mesh.box``: returns (vertices f32[8,3], triangles i32[T,3])."""
    f = np.float32
    dx = np.array([f(length) * f(0.5), 0, 0], dtype=f)
    dy = np.array([0, f(width) * f(0.5), 0], dtype=f)
    dz = np.array([0, 0, f(height) * f(0.5)], dtype=f)
    vertices = np.stack(
        (
            +dx + dy + dz,
            +dx + dy - dz,
            -dx + dy - dz,
            -dx + dy + dz,
            -dx - dy - dz,
            -dx - dy + dz,
            +dx - dy - dz,
            +dx - dy + dz,
        )
    ).astype(f)
    tris = [[0, 1, 2], [0, 2, 3], [3, 2, 4], [3, 4, 5], [5, 4, 6], [5, 6, 7], [7, 6, 1], [7, 1, 0]]
    if with_bottom:
        tris += [[1, 4, 2], [1, 6, 4]]
    if with_top:
        tris += [[0, 3, 5], [0, 5, 7]]
    return vertices, np.asarray(tris, dtype=np.int32)


# --------------------------------------------------------------------------------------
# ray / triangle operators
# --------------------------------------------------------------------------------------
def ray_intersect_triangle(ray_origins, ray_directions, triangle_vertices, *, epsilon=None):
    """UT:1157-1322 (hard mode), full ``*#batch`` broadcasting.  Returns (t, hit)."""
    o, d, tv = _f32(ray_origins), _f32(ray_directions), _f32(triangle_vertices)
    eps = DEFAULT_EPSILON if epsilon is None else float(epsilon)
    batch = np.broadcast_shapes(o.shape[:-1], d.shape[:-1], tv.shape[:-2])
    o = np.ascontiguousarray(np.broadcast_to(o, (*batch, 3)))
    d = np.ascontiguousarray(np.broadcast_to(d, (*batch, 3)))
    tv = np.ascontiguousarray(np.broadcast_to(tv, (*batch, 3, 3)))
    t = np.empty(batch, dtype=np.float32)
    hit = np.empty(batch, dtype=np.uint8)
    n = int(np.prod(batch, dtype=np.int64))
    if n:
        lib().orc_ray_intersect_triangle_paired(_p(o), _p(d), _p(tv), n, eps, _p(t), _p(hit))
    return t, hit.astype(bool)


def ray_intersect_triangle_dense(o, d, tv, *, epsilon=None):
    """rays [R,3] x triangles [T,3,3] -> (t [R,T], hit [R,T])."""
    o, d, tv = _f32(o), _f32(d), _f32(tv)
    eps = DEFAULT_EPSILON if epsilon is None else float(epsilon)
    R, T = o.shape[0], tv.shape[0]
    t = np.empty((R, T), dtype=np.float32)
    hit = np.empty((R, T), dtype=np.uint8)
    if R and T:
        lib().orc_ray_intersect_triangle_dense(_p(o), _p(d), R, _p(tv), T, eps, _p(t), _p(hit))
    return t, hit.astype(bool)


def _flatten_rays_and_triangles(ray_origins, ray_directions, triangle_vertices, active_triangles):
    o, d, tv = _f32(ray_origins), _f32(ray_directions), _f32(triangle_vertices)
    T = tv.shape[-3]
    act = None if active_triangles is None else np.asarray(active_triangles, dtype=bool)
    batch = np.broadcast_shapes(
        o.shape[:-1], d.shape[:-1], tv.shape[:-3], act.shape[:-1] if act is not None else ()
    )
    R = int(np.prod(batch, dtype=np.int64))
    o = np.ascontiguousarray(np.broadcast_to(o, (*batch, 3))).reshape(R, 3)
    d = np.ascontiguousarray(np.broadcast_to(d, (*batch, 3))).reshape(R, 3)
    shared_tv = all(s == 1 for s in tv.shape[:-3])
    if shared_tv:
        tvf, tv_stride = np.ascontiguousarray(tv.reshape(T, 3, 3)), 0
    else:
        tvf = np.ascontiguousarray(np.broadcast_to(tv, (*batch, T, 3, 3))).reshape(R, T, 3, 3)
        tv_stride = 9 * T
    if act is None:
        actf, act_stride = None, 0
    elif all(s == 1 for s in act.shape[:-1]):
        actf, act_stride = np.ascontiguousarray(act.reshape(T)).astype(np.uint8), 0
    else:
        actf = np.ascontiguousarray(np.broadcast_to(act, (*batch, T))).reshape(R, T)
        actf, act_stride = actf.astype(np.uint8), T
    return batch, R, T, o, d, tvf, tv_stride, actf, act_stride


def ray_intersect_any_triangle(
    ray_origins,
    ray_directions,
    triangle_vertices,
    active_triangles=None,
    *,
    epsilon=None,
    hit_tol=None,
    batch_size=512,  # noqa: ARG001 - an OR is tile-order independent
):
    """UT:1353-1537 (hard mode)."""
    batch, R, T, o, d, tv, tvs, act, acts = _flatten_rays_and_triangles(
        ray_origins, ray_directions, triangle_vertices, active_triangles
    )
    eps = DEFAULT_EPSILON if epsilon is None else float(epsilon)
    tol = DEFAULT_HIT_TOL if hit_tol is None else float(hit_tol)
    out = np.zeros(R, dtype=np.uint8)
    if R and T:
        lib().orc_ray_intersect_any_triangle(
            _p(o), _p(d), R, _p(tv), T, tvs, _p(act), acts, eps, tol, _p(out)
        )
    return out.astype(bool).reshape(batch)


def first_triangle_hit_by_ray(
    ray_origins,
    ray_directions,
    triangle_vertices,
    active_triangles=None,
    batch_size=512,
    *,
    epsilon=None,
):
    """UT:1775-1960.  Returns (indices int32, t float32); miss = (-1, inf)."""
    batch, R, T, o, d, tv, tvs, act, acts = _flatten_rays_and_triangles(
        ray_origins, ray_directions, triangle_vertices, active_triangles
    )
    eps = DEFAULT_EPSILON if epsilon is None else float(epsilon)
    idx = np.full(R, -1, dtype=np.int32)
    t = np.full(R, np.inf, dtype=np.float32)
    if R:
        lib().orc_first_triangle_hit_by_ray(
            _p(o), _p(d), R, _p(tv), T, tvs, _p(act), acts, eps,
            0 if batch_size is None else int(batch_size), _p(idx), _p(t),
        )
    return idx.reshape(batch), t.reshape(batch)


# --------------------------------------------------------------------------------------
# image method
# --------------------------------------------------------------------------------------
def image_of_vertex_with_respect_to_mirror(vertex, mirror_vertex, mirror_normal):
    """IM:11-79."""
    x, p, n = _f32(vertex), _f32(mirror_vertex), _f32(mirror_normal)
    batch = np.broadcast_shapes(x.shape[:-1], p.shape[:-1], n.shape[:-1])
    x, p, n = (np.ascontiguousarray(np.broadcast_to(a, (*batch, 3))) for a in (x, p, n))
    out = np.empty((*batch, 3), dtype=np.float32)
    cnt = int(np.prod(batch, dtype=np.int64))
    if cnt:
        lib().orc_image_of_vertex(_p(x), _p(p), _p(n), cnt, _p(out))
    return out


def intersection_of_ray_with_plane(ray_origin, ray_direction, plane_vertex, plane_normal):
    """IM:82-135."""
    o, d, p, n = (_f32(a) for a in (ray_origin, ray_direction, plane_vertex, plane_normal))
    batch = np.broadcast_shapes(o.shape[:-1], d.shape[:-1], p.shape[:-1], n.shape[:-1])
    o, d, p, n = (np.ascontiguousarray(np.broadcast_to(a, (*batch, 3))) for a in (o, d, p, n))
    out = np.empty((*batch, 3), dtype=np.float32)
    cnt = int(np.prod(batch, dtype=np.int64))
    if cnt:
        lib().orc_intersection_of_ray_with_plane(_p(o), _p(d), _p(p), _p(n), cnt, _p(out))
    return out


def image_method(from_vertex, to_vertex, mirror_vertices, mirror_normals):
    """IM:206-363."""
    a, b = _f32(from_vertex), _f32(to_vertex)
    mv, mn = _f32(mirror_vertices), _f32(mirror_normals)
    k = mv.shape[-2]
    batch = np.broadcast_shapes(a.shape[:-1], b.shape[:-1], mv.shape[:-2], mn.shape[:-2])
    out = np.empty((*batch, k, 3), dtype=np.float32)
    B = int(np.prod(batch, dtype=np.int64))
    if k == 0 or B == 0:
        return out
    a = np.ascontiguousarray(np.broadcast_to(a, (*batch, 3)))
    b = np.ascontiguousarray(np.broadcast_to(b, (*batch, 3)))
    mv = np.ascontiguousarray(np.broadcast_to(mv, (*batch, k, 3)))
    mn = np.ascontiguousarray(np.broadcast_to(mn, (*batch, k, 3)))
    lib().orc_image_method(_p(a), _p(b), _p(mv), _p(mn), B, k, _p(out))
    return out


def assemble_path(from_vertex, intermediate_vertices, to_vertex):
    """UT:514-565."""
    a, m, b = _f32(from_vertex), _f32(intermediate_vertices), _f32(to_vertex)
    batch = np.broadcast_shapes(a.shape[:-1], m.shape[:-2], b.shape[:-1])
    return np.concatenate(
        (
            np.broadcast_to(a[..., None, :], (*batch, 1, 3)),
            np.broadcast_to(m, (*batch, *m.shape[-2:])),
            np.broadcast_to(b[..., None, :], (*batch, 1, 3)),
        ),
        axis=-2,
    )


def consecutive_vertices_are_on_same_side_of_mirror(vertices, mirror_vertices, mirror_normals):
    """IM:386-454 (hard mode)."""
    v, mv, mn = _f32(vertices), _f32(mirror_vertices), _f32(mirror_normals)
    k = mv.shape[-2]
    if v.shape[-2] != k + 2:  # IM:422-424
        raise TypeError(f"expected {k + 2} vertices, got {v.shape[-2]}")
    batch = np.broadcast_shapes(v.shape[:-2], mv.shape[:-2], mn.shape[:-2])
    out = np.empty((*batch, k), dtype=np.uint8)
    B = int(np.prod(batch, dtype=np.int64))
    if k and B:
        v = np.ascontiguousarray(np.broadcast_to(v, (*batch, k + 2, 3)))
        mv = np.ascontiguousarray(np.broadcast_to(mv, (*batch, k, 3)))
        mn = np.ascontiguousarray(np.broadcast_to(mn, (*batch, k, 3)))
        lib().orc_same_side_of_mirror(_p(v), _p(mv), _p(mn), B, k, _p(out))
    return out.astype(bool)


# --------------------------------------------------------------------------------------
# fused trace (SV:499-770)
# --------------------------------------------------------------------------------------
def trace_path_candidates(
    vertices,
    triangles,
    tx,
    rx,
    path_candidates,
    *,
    mask=None,
    assume_quads=False,
    epsilon=None,
    hit_tol=None,
    min_len=None,
    return_diag=False,
):
    """SV:499-770, hard mode, dense layout.

    Returns dict(vertices [Ntx,Nrx,C,k+2,3], objects [Ntx,Nrx,C,k+2] int32, mask [Ntx,Nrx,C]).
    """
    tvs = triangle_vertices(vertices, triangles)
    nrm = mesh_normals(tvs)
    T = tvs.shape[0]
    txa, rxa = _f32(tx).reshape(-1, 3), _f32(rx).reshape(-1, 3)
    cand = np.ascontiguousarray(np.asarray(path_candidates, dtype=np.int32))
    Cn, k = cand.shape
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    pr = _TraceParams(
        DEFAULT_EPSILON if epsilon is None else float(epsilon),
        DEFAULT_HIT_TOL if hit_tol is None else float(hit_tol),
        DEFAULT_MIN_LEN if min_len is None else float(min_len),
        1 if assume_quads else 0,
    )
    Ntx, Nrx = txa.shape[0], rxa.shape[0]
    verts = np.zeros((Ntx, Nrx, Cn, k + 2, 3), dtype=np.float32)
    objs = np.zeros((Ntx, Nrx, Cn, k + 2), dtype=np.int32)
    msk = np.zeros((Ntx, Nrx, Cn), dtype=np.uint8)
    diag = np.zeros((Ntx, Nrx, Cn), dtype=np.uint8)
    if Ntx * Nrx * Cn:
        rc = lib().orc_trace_path_candidates(
            _p(tvs), _p(nrm), _p(m), T, _p(txa), Ntx, _p(rxa), Nrx, _p(cand), Cn, k,
            C.byref(pr), _p(verts), _p(objs), _p(msk), _p(diag),
        )
        if rc != 0:
            raise ValueError("order too large for the oracle")
    out = {"vertices": verts, "objects": objs, "mask": msk.astype(bool)}
    if return_diag:
        out["diag"] = diag
    return out


# --------------------------------------------------------------------------------------
# candidate enumeration (GR)
# --------------------------------------------------------------------------------------
class CompleteGraphIter:
    """GR:286-491 ``AllPathsFromCompleteGraphIter`` (literal odometer)."""

    def __init__(self, num_nodes, from_, to, depth, include_from_and_to=True):
        self._L = lib()
        self._h = self._L.orc_cg_iter_new(num_nodes, from_, to, depth, int(include_from_and_to))
        self._buf = np.zeros(depth + 2, dtype=np.uint64)
        self.path_depth = int(self._L.orc_cg_iter_path_depth(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_cg_iter_free(self._h)
            self._h = None

    def __len__(self):
        return min(self.remaining, (1 << 63) - 1)  # Python caps len() at ssize_t

    @property
    def remaining(self) -> int:
        return int(self._L.orc_cg_iter_len(self._h))

    @property
    def overflowed(self) -> bool:
        return bool(self._L.orc_cg_iter_overflowed(self._h))

    def __iter__(self):
        return self

    def __next__(self):
        if not self._L.orc_cg_iter_next(self._h, _p(self._buf)):
            raise StopIteration
        return self._buf[: self.path_depth].astype(np.int64)

    def collect_array(self, max_paths=None):
        n = self.remaining if max_paths is None else int(max_paths)
        if n <= (1 << 24):
            out = np.zeros((n, self.path_depth), dtype=np.uint64)
            got = self._L.orc_cg_iter_collect(self._h, _p(out), n)
            return out[:got].astype(np.int64)
        # declared length is huge (or the reference's overflow sentinel): iterate like collect_array
        # of graph.rs:40-62 does, growing as we go
        chunks, total = [], 0
        while total < n:
            buf = np.zeros((1 << 16, self.path_depth), dtype=np.uint64)
            got = int(self._L.orc_cg_iter_collect(self._h, _p(buf), 1 << 16))
            if got == 0:
                break
            chunks.append(buf[:got].astype(np.int64))
            total += got
        if not chunks:
            return np.zeros((0, self.path_depth), dtype=np.int64)
        return np.concatenate(chunks)


def generate_all_path_candidates(num_primitives: int, order: int) -> np.ndarray:
    """UT:1047-1081."""
    it = CompleteGraphIter(num_primitives, num_primitives, num_primitives + 1, order + 2, False)
    return it.collect_array()


class DiGraph:
    """GR:594-1120 ``DiGraph`` (adjacency lists + DFS path iterator)."""

    def __init__(self, handle):
        self._L = lib()
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_digraph_free(self._h)
            self._h = None

    @classmethod
    def from_complete_graph(cls, num_nodes: int) -> "DiGraph":
        return cls(lib().orc_digraph_from_complete_graph(num_nodes))

    @classmethod
    def from_adjacency_matrix(cls, m) -> "DiGraph":
        m = np.ascontiguousarray(np.asarray(m, dtype=np.uint8))
        return cls(lib().orc_digraph_from_adjacency_matrix(_p(m), m.shape[0]))

    @property
    def num_nodes(self) -> int:
        return int(self._L.orc_digraph_num_nodes(self._h))

    def insert_from_and_to_nodes(self, *, direct_path=True, from_adjacency=None, to_adjacency=None):
        fa = None if from_adjacency is None else np.ascontiguousarray(from_adjacency, dtype=np.uint8)
        ta = None if to_adjacency is None else np.ascontiguousarray(to_adjacency, dtype=np.uint8)
        f, t = C.c_uint64(), C.c_uint64()
        self._L.orc_digraph_insert_from_and_to_nodes(
            self._h, int(direct_path), _p(fa), _p(ta), C.byref(f), C.byref(t)
        )
        return int(f.value), int(t.value)

    def filter_by_mask(self, mask, fast_mode=True):
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        if self._L.orc_digraph_filter_by_mask(self._h, _p(m), m.shape[0], int(fast_mode)) != 0:
            raise ValueError("'mask' length must be smaller than or equal to the number of nodes")

    def disconnect_nodes(self, *nodes, fast_mode=True):
        n = np.asarray(nodes, dtype=np.uint64)
        if self._L.orc_digraph_disconnect_nodes(self._h, _p(n), n.shape[0], int(fast_mode)) != 0:
            raise IndexError("node out-of-bounds")

    def all_paths_array(self, from_, to, depth, *, include_from_and_to=True, max_paths=1 << 24):
        it = self._L.orc_dg_iter_new(self._h, from_, to, depth, int(include_from_and_to))
        try:
            pd = int(self._L.orc_dg_iter_path_depth(it))
            chunks, buf = [], np.zeros((4096, max(pd, 1)), dtype=np.uint64)
            total = 0
            while total < max_paths:
                got = int(self._L.orc_dg_iter_collect(it, _p(buf), 4096)) if pd else 0
                if pd == 0:
                    # depth-2 paths carry no intermediate node: count them one by one
                    tmp = np.zeros(depth + 2, dtype=np.uint64)
                    cnt = 0
                    while self._L.orc_dg_iter_next(it, _p(tmp)):
                        cnt += 1
                    return np.zeros((cnt, 0), dtype=np.int64)
                if got == 0:
                    break
                chunks.append(buf[:got, :pd].astype(np.int64))
                total += got
            if not chunks:
                return np.zeros((0, pd), dtype=np.int64)
            return np.concatenate(chunks, axis=0)
        finally:
            self._L.orc_dg_iter_free(it)


def differentiable_distance(tv, o, d, faces):
    """ME:226-255."""
    tv, o, d = _f32(tv), _f32(o), _f32(d)
    f = np.ascontiguousarray(np.asarray(faces, dtype=np.int32))
    out = np.empty(f.shape[0], dtype=np.float32)
    if f.shape[0]:
        lib().orc_differentiable_distance(_p(tv), _p(o), _p(d), _p(f), f.shape[0], _p(out))
    return out


# --------------------------------------------------------------------------------------
# visibility by ray launching ("next" row f2): UT:369-490, 639-993, 1540-1772 (NumPy float32)
# Transcendentals (cos/sin/arccos/arctan2) are NumPy's float32 routines: XLA's may differ in the
# last ulp, so lattice directions are pinned at ~1e-7 relative, visible SETS by the reference's
# known-answer tests (test_utils.py:717-827).
# --------------------------------------------------------------------------------------
def cartesian_to_spherical(xyz):
    """UT:930-958."""
    xyz = _f32(xyz)
    r = np.sqrt((xyz * xyz).sum(-1, dtype=np.float32)).astype(np.float32)
    r = np.where(r == 0.0, np.float32(1.0), r).astype(np.float32)
    p = np.arccos((xyz[..., 2] / r).astype(np.float32)).astype(np.float32)
    a = np.arctan2(xyz[..., 1], xyz[..., 0]).astype(np.float32)
    return np.stack((r, p, a), axis=-1)


def spherical_to_cartesian(rpa):
    """UT:961-993."""
    rpa = _f32(rpa)
    p, a = rpa[..., -2], rpa[..., -1]
    cp, sp, ca, sa = np.cos(p), np.sin(p), np.cos(a), np.sin(a)
    xyz = np.stack((sp * ca, sp * sa, cp), axis=-1).astype(np.float32)
    if rpa.shape[-1] == 3:
        xyz = xyz * rpa[..., 0, None]
    return xyz.astype(np.float32)


def viewing_frustum(viewing_vertex, world_vertices, *, active_vertices=None):
    """UT:639-927 (reduce=False): [*batch, 2, 3] = [[r_min, p_min, a_min], [r_max, p_max, a_max]]."""
    f = np.float32
    wv, vv = _f32(world_vertices), _f32(viewing_vertex)
    rpa = cartesian_to_spherical(wv - vv[..., None, :])
    r, p, a = rpa[..., 0], rpa[..., 1], rpa[..., 2]
    act = None if active_vertices is None else np.broadcast_to(np.asarray(active_vertices, bool), r.shape)

    def red(fn, x, init):
        if act is None:
            return fn(np.concatenate((x, np.full((*x.shape[:-1], 1), init, f)), -1), axis=-1).astype(f)
        return fn(np.where(act, x, f(init)), axis=-1, initial=f(init)).astype(f)

    pi, two_pi = f(np.pi), f(2 * np.pi)
    r_min, r_max = red(np.min, r, np.inf), red(np.max, r, 0)
    p_min, p_max = red(np.min, p, pi), red(np.max, p, 0)
    a_min, a_max = red(np.min, a, pi), red(np.max, a, -pi)
    a_0 = np.mod(a + two_pi, two_pi).astype(f)
    a_0_min, a_0_max = red(np.min, a_0, two_pi), red(np.max, a_0, 0)
    a_width, a_0_width = a_max - a_min, a_0_max - a_0_min
    swap = a_width > a_0_width
    a_min, a_max = np.where(swap, a_0_min, a_min), np.where(swap, a_0_max, a_max)
    full = np.minimum(a_width, a_0_width) > f(1.5) * pi
    a_min, a_max = np.where(full, -pi, a_min), np.where(full, pi, a_max)
    p_0_min, p_0_max = p_min, p_max
    deg = p_min == p_max
    p_min = np.where(deg, f(0.0), p_min)
    p_0_max = np.where(deg, pi, p_0_max)
    sw = (p_max - p_min) > (p_0_max - p_0_min)
    p_min, p_max = np.where(sw, p_0_min, p_min), np.where(sw, p_0_max, p_max)
    out = np.stack((r_min, p_min, a_min, r_max, p_max, a_max), axis=-1).astype(f)
    return out.reshape(*r.shape[:-1], 2, 3)


def fibonacci_lattice(n: int, frustum=None):
    """UT:369-490 (float32), incl. the split-modulus evaluation of frac(i / phi) (:426-462)."""
    if n <= 0:
        raise ValueError(f"Invalid size {n!r}, must be strictly positive.")
    f = np.float32
    i = np.arange(0.0, n, dtype=f)
    inv_phi, m1, m2 = 0.6180339887498949, 262144.0, 512.0
    inv_phi_m1, inv_phi_m2 = f((inv_phi * m1) % 1.0), f((inv_phi * m2) % 1.0)
    q1 = np.floor(i / f(m1)).astype(f)
    rem = (i - q1 * f(m1)).astype(f)
    q2 = np.floor(rem / f(m2)).astype(f)
    r = (rem - q2 * f(m2)).astype(f)
    frac = np.mod(((q1 * inv_phi_m1).astype(f) + (q2 * inv_phi_m2).astype(f)).astype(f) + (r * f(inv_phi)).astype(f), f(1.0)).astype(f)
    if frustum is not None:
        fr = _f32(frustum)
        p_min, a_min, p_max, a_max = fr[0, -2], fr[0, -1], fr[1, -2], fr[1, -1]
        cos_p_min, cos_p_max = np.cos(p_min), np.cos(p_max)
        denom = f(n - 1) if n > 1 else f(1.0)
        cos_lat = (cos_p_min - ((cos_p_min - cos_p_max) * (i / denom).astype(f)).astype(f)).astype(f)
        lat = np.arccos(cos_lat).astype(f)
        lon = (a_min + ((a_max - a_min) * frac).astype(f)).astype(f)
    else:
        lat = np.arccos((f(1) - (f(2) * i).astype(f) / f(n)).astype(f)).astype(f)
        lon = ((f(2) * f(np.pi)) * frac).astype(f)
    return spherical_to_cartesian(np.stack((lat, lon), axis=-1))


def triangles_visible_from_vertex(vertex, triangle_vertices_, active_triangles=None, num_rays=int(1e6),
                                  *, epsilon=None):
    """UT:1540-1772: frustum -> Fibonacci lattice rays -> first hit (one tile) -> scatter."""
    v = _f32(vertex)
    tv = _f32(triangle_vertices_)
    T = tv.shape[0]
    batch = v.shape[:-1]
    vf = v.reshape(-1, 3)
    out = np.zeros((vf.shape[0], T), dtype=bool)
    if T == 0:
        return out.reshape(*batch, T)
    centers = tv.mean(axis=-2, keepdims=True, dtype=np.float32)
    world = np.concatenate((tv, centers), axis=-2).reshape(-1, 3)
    act = None if active_triangles is None else np.asarray(active_triangles, bool)
    actv = None if act is None else np.repeat(act, 4, axis=-1)
    for b in range(vf.shape[0]):
        fr = viewing_frustum(vf[b], world, active_vertices=actv)
        dirs = fibonacci_lattice(num_rays, frustum=fr)
        idx, _ = first_triangle_hit_by_ray(np.broadcast_to(vf[b], dirs.shape), dirs, tv, act,
                                           batch_size=None, epsilon=epsilon)
        out[b, idx[idx >= 0]] = True
    return out.reshape(*batch, T)


# --------------------------------------------------------------------------------------
# shooting-and-bouncing rays ("next" row f3): SV:279-491, 1179-1226 (NumPy float32 + C first-hit)
# --------------------------------------------------------------------------------------
def _dot32(a, b):
    p = (a * b).astype(np.float32)
    return ((p[..., 0] + p[..., 1]).astype(np.float32) + p[..., 2]).astype(np.float32)


def _cross32(a, b):
    f = np.float32
    x = ((a[..., 1] * b[..., 2]).astype(f) - (a[..., 2] * b[..., 1]).astype(f)).astype(f)
    y = ((a[..., 2] * b[..., 0]).astype(f) - (a[..., 0] * b[..., 2]).astype(f)).astype(f)
    z = ((a[..., 0] * b[..., 1]).astype(f) - (a[..., 1] * b[..., 0]).astype(f)).astype(f)
    return np.stack((x, y, z), axis=-1)


def sbr_launch_rays(vertices, triangles, tx, rx, num_rays):
    """SBRPathLauncher.launch_rays SV:1202-1226: frustum over triangle vertices + receivers, lattice."""
    tv = triangle_vertices(vertices, triangles)
    txa, rxa = _f32(tx).reshape(-1, 3), _f32(rx).reshape(-1, 3)
    world = np.concatenate((tv.reshape(-1, 3), rxa), axis=0)
    dirs = np.stack([fibonacci_lattice(num_rays, frustum=viewing_frustum(t, world)) for t in txa])
    return np.broadcast_to(txa[:, None, :], dirs.shape).copy(), dirs


def launch_paths(vertices, triangles, ray_origins, ray_directions, rx, order, *, mask=None,
                 max_dist=1e-3, epsilon=None, batch_size=512):
    """AbstractPathLauncher.launch_paths SV:358-491 for given rays [Ntx, R, 3] (first hit = the pure
    operator UT:1775-1960; the reference dispatches to Warp here).  Returns dict(triangles
    [Ntx,R,order], vertices [Ntx,R,order,3], masks [Ntx,Nrx,R,order+1])."""
    f = np.float32
    tv = triangle_vertices(vertices, triangles)
    nrm = mesh_normals(tv)
    o, d = _f32(ray_origins).copy(), _f32(ray_directions).copy()
    rxa = _f32(rx).reshape(-1, 3)
    ntx, R = o.shape[0], o.shape[1]
    valid = np.ones((ntx, R), bool)
    tris, verts, masks = [], [], []
    md = f(max_dist)
    for _ in range(order + 1):
        idx, t_hit = first_triangle_hit_by_ray(o, d, tv, mask, batch_size=batch_size, epsilon=epsilon)
        v = (rxa[None, :, None, :] - o[:, None, :, :]).astype(f)          # SV:340-342
        c = _cross32(np.broadcast_to(d[:, None], v.shape), v)
        dist2 = (((c[..., 0] * c[..., 0]).astype(f) + (c[..., 1] * c[..., 1]).astype(f)).astype(f)
                 + (c[..., 2] * c[..., 2]).astype(f)).astype(f)            # SV:343-345
        t_rx = _dot32(np.broadcast_to(d[:, None], v.shape), v)            # SV:346-348
        masks.append((t_rx > 0) & (t_rx < t_hit[:, None, :]) & valid[:, None, :] & (dist2 < md))
        inside = np.isfinite(t_hit)                                        # SV:300-303
        valid = valid & inside
        t = np.where(inside, t_hit, f(0)).astype(f)
        o = (o + (t[..., None] * d).astype(f)).astype(f)                   # SV:305
        n = nrm[np.where(idx >= 0, idx, len(nrm) - 1)] if len(nrm) else np.zeros_like(d)
        c2 = (f(2.0) * _dot32(d, n)).astype(f)
        d = (d - (c2[..., None] * n).astype(f)).astype(f)                  # SV:307-312
        tris.append(idx)
        verts.append(o.copy())
    return {
        "triangles": np.stack(tris[:-1], axis=-1) if order else np.zeros((ntx, R, 0), np.int32),
        "vertices": np.stack(verts[:-1], axis=-2) if order else np.zeros((ntx, R, 0, 3), f),
        "masks": np.stack(masks, axis=-1),
    }


# --------------------------------------------------------------------------------------
# cell ids of equal rows (PA:21-38 `_cell_ids`: a reverse scan comparing every row with every other)
# --------------------------------------------------------------------------------------
def cell_ids(rows) -> np.ndarray:
    """``out[r]`` = smallest ``i`` with ``rows[i] == rows[r]`` (the reverse scan of PA:24-38 leaves,
    for every row, the LAST index written = the smallest matching one)."""
    rows = np.asarray(rows)
    n = rows.shape[0]
    if n <= 2048:  # literal restatement
        out = np.empty(n, dtype=np.int32)
        for index in range(n - 1, -1, -1):
            out[(rows == rows[index]).all(axis=-1)] = index
        return out
    _, first, inverse = np.unique(rows, axis=0, return_index=True, return_inverse=True)
    return first[inverse.reshape(-1)].astype(np.int32)
