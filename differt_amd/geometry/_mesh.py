"""``Mesh``: the part of the reference's triangle mesh that sits on the hot path.

Mirrors the fields and the mesh-bound ray queries of ``differt/src/differt/geometry/_mesh.py``
(fields :624-688, ``triangle_vertices`` :899-905, ``normals`` :950-956, ``box`` :2109-2217,
``ray_intersect_any_triangle`` :3018-3094, ``first_triangle_hit_by_ray`` :3096-3162).  Scene
authoring (sampling, editing, plotting, file IO) is out of scope.
"""

from __future__ import annotations

import ctypes as C
import itertools
from dataclasses import dataclass, field, replace

import numpy as np
import torch

from .. import _lib
from .._tensors import F32_EPS, as_f32, as_i32, device, ptr, stream
from . import _utils

_GENERATIONS = itertools.count(1)  # Mesh.generation(): one number per device snapshot, never reused

__all__ = ["Mesh"]


class _MeshHandle:
    """Owns one ``drt_mesh_t`` (explicit replacement of the reference's ``_WARP_MESHES_CACHE``)."""

    def __init__(self, vertices, triangles, mask, assume_quads: bool):
        h = C.c_void_p()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        _lib.call(
            "drt_mesh_create", ptr(vertices), vertices.shape[0], ptr(triangles), triangles.shape[0],
            ptr(m), int(assume_quads), stream(), C.byref(h),
        )
        self.h = h
        self.num_triangles = triangles.shape[0]

    def __del__(self):
        if getattr(self, "h", None):
            try:
                _lib.load().drt_mesh_destroy(self.h)
            except Exception:  # noqa: BLE001 - interpreter shutdown: modules may already be torn down
                pass
            self.h = None

    def triangle_vertices(self) -> torch.Tensor:
        out = torch.empty((self.num_triangles, 3, 3), dtype=torch.float32, device=device())
        _lib.call("drt_mesh_copy", self.h, ptr(out), None, stream())
        return out

    def normals(self) -> torch.Tensor:
        out = torch.empty((self.num_triangles, 3), dtype=torch.float32, device=device())
        _lib.call("drt_mesh_copy", self.h, None, ptr(out), stream())
        return out


class _FirstHitFn(torch.autograd.Function):
    """``t`` of the first hit, differentiable in (vertices, origins, directions) like the
    reference's custom VJP (_mesh.py:258-344): the backward pass re-evaluates Moller-Trumbore on the
    hit face only; indices carry no gradient."""

    @staticmethod
    def forward(ctx, vertices, origins, directions, triangles, mask, epsilon, batch_size, bvh_handle=None,
                nudge=0.0):
        R = origins.shape[0]
        idx = torch.full((R,), -1, dtype=torch.int32, device=origins.device)
        t = torch.full((R,), float("inf"), dtype=torch.float32, device=origins.device)
        qo, qd = origins, directions
        if R and nudge:  # Warp semantics (_mesh.py:195-199): query from origin + 1e-5 * direction
            qo, qd = torch.empty_like(origins), torch.empty_like(directions)
            _lib.call("drt_warp_ray_prep", ptr(origins), ptr(directions), R, 1, float(nudge), ptr(qo), ptr(qd), stream())
        if R and bvh_handle is not None:
            _lib.call("drt_mesh_first_triangle_hit_by_ray", bvh_handle.h, ptr(qo), ptr(qd), R,
                      epsilon, batch_size, ptr(idx), ptr(t), stream())
        elif R:
            tv = vertices[triangles.long()].contiguous()
            ws = torch.empty(R, dtype=torch.int64, device=origins.device)
            m = None if mask is None else mask.to(torch.uint8).contiguous()
            _lib.call(
                "drt_first_triangle_hit_by_ray", ptr(qo), ptr(qd), R, ptr(tv),
                tv.shape[0], 0, ptr(m), 0, epsilon, batch_size, ptr(idx), ptr(t), ptr(ws), R * 8,
                stream(),
            )
        if R and nudge:
            _lib.call("drt_warp_first_hit_finish", ptr(idx), ptr(t), R, float(nudge), stream())
        # the backward pass differentiates the UN-nudged distance on the hit face, like the reference's
        # custom VJP (_mesh.py:327-338)
        ctx.save_for_backward(vertices, origins, directions, triangles, idx)
        ctx.mark_non_differentiable(idx)
        return idx, t

    @staticmethod
    def backward(ctx, _gidx, gt):
        vertices, origins, directions, triangles, idx = ctx.saved_tensors
        R = origins.shape[0]
        gv = torch.zeros_like(vertices)
        go, gd = torch.zeros_like(origins), torch.zeros_like(directions)
        if R:
            _lib.call(
                "drt_first_hit_vjp", ptr(vertices), ptr(triangles), ptr(origins), ptr(directions),
                ptr(idx), ptr(gt.contiguous()), R, ptr(gv), ptr(go), ptr(gd), stream(),
            )
        return gv, go, gd, None, None, None, None, None, None


@dataclass
class Mesh:
    """Triangle mesh in HBM: ``vertices f32[Nv,3]``, ``triangles i32[T,3]``, optional
    ``mask bool[T]`` (inactive triangles neither reflect nor occlude) and ``assume_quads``
    (consecutive triangle pairs form planar quads, _mesh.py:851-886)."""

    vertices: torch.Tensor
    triangles: torch.Tensor
    mask: torch.Tensor | None = None
    assume_quads: bool = False
    object_bounds: torch.Tensor | None = None
    face_materials: torch.Tensor | None = None
    """``i32[T]`` index into ``material_names`` (reference fields _mesh.py:661-673)."""
    material_names: tuple[str, ...] = ()
    _handle: _MeshHandle | None = field(default=None, repr=False, compare=False)

    def __post_init__(self):
        dev = device()
        self.vertices = as_f32(self.vertices, dev).reshape(-1, 3)
        self.triangles = as_i32(self.triangles, dev).reshape(-1, 3).contiguous()
        if self.mask is not None:
            m = self.mask if isinstance(self.mask, torch.Tensor) else torch.as_tensor(np.asarray(self.mask))
            self.mask = m.to(device=dev).bool().reshape(-1).contiguous()
            if self.mask.shape[0] != self.triangles.shape[0]:
                raise ValueError("mask must have one entry per triangle")
        if self.assume_quads and self.triangles.shape[0] % 2 != 0:
            raise ValueError("assume_quads requires an even number of triangles")  # _mesh.py:690-696

    # ---- native handle: a device snapshot of (vertices, triangles, mask), keyed on the identity AND
    # the in-place version of the tensors it was taken from, so that `optimizer.step()` on
    # `mesh.vertices`, `mesh.vertices -= ...` or `mesh.mask = ...` can never leave the kernels (trace,
    # normals, VJP) on stale geometry while `triangle_vertices` reads the live tensor ----
    def _handle_key(self) -> tuple:
        def k(t):
            return None if t is None else (t.data_ptr(), t._version, tuple(t.shape))

        return (k(self.vertices), k(self.triangles), k(self.mask), bool(self.assume_quads))

    def _handle_is_current(self) -> bool:
        """The handle keeps STRONG references to the tensors it was snapshotted from and is current only
        while the mesh still holds those very objects (``is``) at the same in-place version.  An address /
        version pair alone is not an identity: a fresh tensor from an optimisation step can be allocated
        where a freed one lived, with version 0 again."""
        h = self._handle
        if h is None or getattr(h, "key", None) != self._handle_key():
            return False
        return all(a is b for a, b in zip(h.src, (self.vertices, self.triangles, self.mask)))

    def handle(self) -> _MeshHandle:
        if not self._handle_is_current():
            if self.mask is not None and self.mask.shape[0] != self.triangles.shape[0]:
                raise ValueError("mask must have one entry per triangle")
            self._handle = _MeshHandle(
                self.vertices.detach().contiguous(), self.triangles, self.mask, self.assume_quads
            )
            self._handle.key = self._handle_key()
            self._handle.src = (self.vertices, self.triangles, self.mask)
            self._handle.generation = next(_GENERATIONS)
        return self._handle

    def generation(self) -> int:
        """Identity of the current device snapshot (changes whenever :meth:`handle` re-snapshots): the key
        for anything cached per geometry outside the native handle.  A process-wide counter, not ``id()``: the address
        of a freed handle can come back with the next snapshot."""
        return self.handle().generation

    def beam_pairing(self) -> dict:
        """What the pruned tracer's pairing pass makes of this (triangle) mesh (``drt_mesh_build_beam_clusters`` +
        ``drt_mesh_beam_pairing``; built on first use, cached in the native handle): ``pair_mode`` -- the search runs
        over ``primitives`` < ``num_triangles`` primitives, ``pairs`` of them coplanar triangle pairs (two triangles
        ``(v0, v1, v2)``, ``(v0, v2, v3)`` anywhere in the mesh that are the same mirror for the reference, convex
        union) -- or triangle by triangle (too few pairs, or ``assume_quads``, whose quads are the primitives)."""
        h = self.handle().h
        if self.triangles.shape[0]:
            _lib.call("drt_mesh_build_beam_clusters", h, stream())
        prims, pairs = C.c_int64(0), C.c_int64(0)
        state = _lib.load().drt_mesh_beam_pairing(h, C.byref(prims), C.byref(pairs))
        return {"pair_mode": state == 1, "primitives": int(prims.value) if state == 1 else self.num_primitives,
                "pairs": int(pairs.value)}

    # ---- reference properties ----
    @property
    def num_triangles(self) -> int:
        return self.triangles.shape[0]

    @property
    def num_quads(self) -> int:
        if not self.assume_quads:
            raise ValueError("Cannot access the number of quadrilaterals if 'assume_quads' is not set to 'True'.")
        return self.triangles.shape[0] // 2

    @property
    def num_primitives(self) -> int:
        """_mesh.py:879-886."""
        return self.num_quads if self.assume_quads else self.num_triangles

    @property
    def is_empty(self) -> bool:
        return self.triangles.shape[0] == 0

    @property
    def triangle_vertices(self) -> torch.Tensor:
        """``[T,3,3]`` (_mesh.py:899-905); differentiable in ``vertices``."""
        return self.vertices[self.triangles.long()]

    @property
    def normals(self) -> torch.Tensor:
        """``[T,3]`` unit normals computed by the mesh kernel (_mesh.py:950-956)."""
        return self.handle().normals()

    def set_assume_quads(self, flag: bool = True) -> "Mesh":
        return replace(self, assume_quads=flag, _handle=None)

    def set_mask(self, mask) -> "Mesh":
        return replace(self, mask=mask, _handle=None)

    def set_face_materials(self, materials) -> "Mesh":
        """_mesh.py:1977-2003: one index for all triangles or one per triangle (no bounds check)."""
        fm = as_i32(materials).reshape(-1)
        return replace(self, face_materials=fm.expand(self.num_triangles).contiguous())

    def set_materials(self, *names: str) -> "Mesh":
        """_mesh.py:1930-1975: one name for all triangles, one per triangle, or one per quad."""
        if len(names) not in {1, self.num_triangles, self.num_primitives}:
            if self.assume_quads:
                raise ValueError(f"Expected either 1, {self.num_triangles}, or {self.num_primitives} names, "
                                 f"got {len(names)}.")
            raise ValueError(f"Expected either 1, or {self.num_triangles} names, got {len(names)}.")
        table = {name: i for i, name in enumerate(dict.fromkeys((*self.material_names, *names)))}
        idx = np.array([table[name] for name in names], dtype=np.int32)
        if self.assume_quads and len(names) == self.num_quads and len(names) != 1:
            idx = np.repeat(idx, 2)
        return replace(self, material_names=tuple(table)).set_face_materials(idx)

    def with_vertices(self, vertices) -> "Mesh":
        return replace(self, vertices=vertices, _handle=None)

    def masked(self) -> "Mesh":
        """Sub-mesh of the active triangles (_mesh.py ``masked``; used by mask==sub-mesh tests)."""
        if self.mask is None:
            return self
        keep = self.mask
        if self.assume_quads:  # a quad survives as a whole or not at all (_mesh.py:1397-1429)
            pairs = keep.reshape(-1, 2)
            if bool((pairs[:, 0] != pairs[:, 1]).any()):
                raise ValueError("assume_quads: the mask splits a quad (both triangles must share one flag)")
        fm = None if self.face_materials is None else self.face_materials[keep].contiguous()
        return replace(self, triangles=self.triangles[keep].contiguous(), mask=None, _handle=None,
                       object_bounds=None, face_materials=fm)

    def append(self, other: "Mesh") -> "Mesh":
        """Concatenate two meshes (_mesh.py:1555-1734): indices of ``other`` are offset, masks are
        concatenated (a missing mask counts as all-True), ``assume_quads`` = both."""
        tri = torch.cat((self.triangles, other.triangles + self.vertices.shape[0]))
        if self.mask is None and other.mask is None:
            mask = None
        else:
            ma = self.mask if self.mask is not None else torch.ones(self.num_triangles, dtype=torch.bool, device=tri.device)
            mb = other.mask if other.mask is not None else torch.ones(other.num_triangles, dtype=torch.bool, device=tri.device)
            mask = torch.cat((ma, mb))
        # materials (_mesh.py:1571-1575): names merged (unique, self first), other's indices renumbered; a
        # mesh without materials contributes -1 when the other one has some
        names, fm = (), None
        if self.face_materials is not None or other.face_materials is not None:
            table = {n: i for i, n in enumerate(dict.fromkeys((*self.material_names, *other.material_names)))}
            names = tuple(table)

            def remap(m):
                if m.face_materials is None:
                    return torch.full((m.num_triangles,), -1, dtype=torch.int32, device=tri.device)
                lut = torch.as_tensor([table[n] for n in m.material_names] or [0], dtype=torch.int32, device=tri.device)
                f = m.face_materials.to(device=tri.device, dtype=torch.int64)
                ok = (f >= 0) & (f < len(m.material_names))
                return torch.where(ok, lut[f.clamp(0, max(len(m.material_names) - 1, 0))], f.to(torch.int32))

            fm = torch.cat((remap(self), remap(other)))
        return Mesh(torch.cat((self.vertices, other.vertices)), tri, mask,
                    self.assume_quads and other.assume_quads, face_materials=fm, material_names=names)

    __add__ = append

    def translate(self, t) -> "Mesh":
        return replace(self, vertices=self.vertices + as_f32(t), _handle=None)

    @classmethod
    def empty(cls) -> "Mesh":
        return cls(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))

    @classmethod
    def box(cls, length=1.0, width=1.0, height=1.0, *, with_top: bool = False,
            with_bottom: bool = True) -> "Mesh":
        """Vertex/triangle tables of the reference (_mesh.py:2172-2217): 8 vertices, 10 triangles by
        default, 12 with ``with_top``; consecutive triangles pair into quads."""
        f = np.float32
        dx = np.array([f(length) * f(0.5), 0, 0], dtype=f)
        dy = np.array([0, f(width) * f(0.5), 0], dtype=f)
        dz = np.array([0, 0, f(height) * f(0.5)], dtype=f)
        vertices = np.stack((+dx + dy + dz, +dx + dy - dz, -dx + dy - dz, -dx + dy + dz,
                             -dx - dy - dz, -dx - dy + dz, +dx - dy - dz, +dx - dy + dz)).astype(f)
        tris = [[0, 1, 2], [0, 2, 3], [3, 2, 4], [3, 4, 5], [5, 4, 6], [5, 6, 7], [7, 6, 1], [7, 1, 0]]
        if with_bottom:
            tris += [[1, 4, 2], [1, 6, 4]]
        if with_top:
            tris += [[0, 3, 5], [0, 5, 7]]
        tris = np.asarray(tris, dtype=np.int32)
        idx = np.arange(0, tris.shape[0] + 1, 2)
        return cls(vertices, tris, object_bounds=torch.as_tensor(np.column_stack((idx[:-1], idx[1:]))))

    # ---- mesh-bound ray queries ----
    def ray_intersect_any_triangle(self, ray_origins, ray_directions, *, hit_tol: float | None = None,
                                   epsilon: float | None = None, accel: str | None = None,
                                   semantics: str = "jax") -> torch.Tensor:
        """Whether each ray is blocked by an active triangle (_mesh.py:3018-3094; non-differentiable
        like the reference, :3087-3094).  The predicate is the pure-JAX operator's
        (_utils.py:1469); the reference dispatches to a Warp BVH query here, see DESIGN.md.

        ``semantics="warp"`` (opt-in) prepares the rays like the reference's Warp path (:3065-3070):
        unit direction, origin moved ``hit_tol * |d|`` along it, segment shortened to
        ``|d| * (1 - 2 hit_tol)``; the triangle test itself stays this library's Moller-Trumbore (Warp's
        own routine is third-party: parity beyond the reference's equality test is unpinned)."""
        o, d = as_f32(ray_origins), as_f32(ray_directions)
        batch = torch.broadcast_shapes(o.shape[:-1], d.shape[:-1])
        if self.is_empty:  # _mesh.py:3053-3057
            return torch.zeros(batch, dtype=torch.bool, device=o.device)
        if semantics not in ("jax", "warp"):
            raise ValueError(f"unknown semantics {semantics!r}")
        if semantics == "warp":
            tol = 100.0 * F32_EPS if hit_tol is None else float(hit_tol)
            of = o.detach().expand(*batch, 3).contiguous().reshape(-1, 3)
            df = d.detach().expand(*batch, 3).contiguous().reshape(-1, 3)
            o2, d2 = torch.empty_like(of), torch.empty_like(df)
            _lib.call("drt_warp_ray_prep", ptr(of), ptr(df), of.shape[0], 0, tol, ptr(o2), ptr(d2), stream())
            return self.ray_intersect_any_triangle(o2.reshape(*batch, 3), d2.reshape(*batch, 3), hit_tol=0.0,
                                                   epsilon=epsilon, accel=accel)
        if accel == "bvh":  # own LBVH, the counterpart of the reference's Warp BVH (csrc/bvh.hip)
            of = o.detach().expand(*batch, 3).contiguous().reshape(-1, 3)
            df = d.detach().expand(*batch, 3).contiguous().reshape(-1, 3)
            out = torch.empty(of.shape[0], dtype=torch.uint8, device=o.device)
            _lib.call("drt_mesh_ray_intersect_any_triangle", self.handle().h, ptr(of), ptr(df),
                      of.shape[0], 10.0 * F32_EPS if epsilon is None else float(epsilon),
                      100.0 * F32_EPS if hit_tol is None else float(hit_tol), ptr(out), stream())
            return out.bool().reshape(batch)
        if accel is not None:
            raise ValueError(f"unknown accel {accel!r}")
        with torch.no_grad():
            return _utils.ray_intersect_any_triangle(
                o, d, self.handle().triangle_vertices(), self.mask, hit_tol=hit_tol, epsilon=epsilon
            )

    def triangles_visible_from_vertex(self, vertex, num_rays: int = int(1e6), *, accel: str | None = None,
                                      sample_triangles: bool = False) -> torch.Tensor:
        """``bool[*batch, T]``: triangles visible from each vertex (_mesh.py:3164-3253); masked
        triangles are never visible and do not occlude.  ``sample_triangles`` (extension, LBVH only) adds
        the faces that have an unoccluded interior sample point: small far-away faces that fall between the
        lattice rays."""
        v = as_f32(vertex)
        if self.is_empty:
            return torch.zeros((*v.shape[:-1], 0), dtype=torch.bool, device=v.device)
        if accel == "bvh":
            vf = v.detach().reshape(-1, 3).contiguous()
            B, T = vf.shape[0], self.num_triangles
            vis = torch.zeros((B, T), dtype=torch.uint8, device=v.device)
            ws = torch.empty((max(B, 1), 6), dtype=torch.float32, device=v.device)
            if B:
                _lib.call("drt_mesh_triangles_visible_from_vertex", self.handle().h, ptr(vf), B, int(num_rays),
                          10.0 * F32_EPS, ptr(vis), ptr(ws), stream())
                if sample_triangles:
                    _lib.call("drt_mesh_triangles_visible_samples", self.handle().h, ptr(vf), B, 10.0 * F32_EPS,
                              ptr(vis), stream())
            return vis.bool().reshape(*v.shape[:-1], T)
        if accel is not None:
            raise ValueError(f"unknown accel {accel!r}")
        if sample_triangles:
            raise ValueError("sample_triangles needs accel='bvh'")
        with torch.no_grad():
            return _utils.triangles_visible_from_vertex(v, self.handle().triangle_vertices(), self.mask,
                                                        num_rays=num_rays)

    def first_triangle_hit_by_ray(self, ray_origins, ray_directions, *, epsilon: float | None = None,
                                  batch_size: int | None = 512, accel: str | None = None,
                                  semantics: str = "jax"):
        """Closest hit ``(index, t)``; ``t`` is differentiable w.r.t. origins, directions and mesh
        vertices (_mesh.py:3096-3162, custom VJP :258-344).  Miss = ``(-1, inf)``.

        ``semantics="warp"`` (opt-in): query from ``origin + 1e-5 * direction`` and add ``1e-5`` back to
        ``t`` like the reference's Warp kernel (_mesh.py:195-199); gradients unchanged (:327-338)."""
        o, d = as_f32(ray_origins), as_f32(ray_directions)
        batch = torch.broadcast_shapes(o.shape[:-1], d.shape[:-1])
        if self.is_empty:  # _mesh.py:3129-3136
            return (torch.full(batch, -1, dtype=torch.int32, device=o.device),
                    torch.full(batch, float("inf"), dtype=torch.float32, device=o.device))
        of = o.expand(*batch, 3).contiguous().reshape(-1, 3)
        df = d.expand(*batch, 3).contiguous().reshape(-1, 3)
        eps = 10.0 * F32_EPS if epsilon is None else float(epsilon)
        if accel not in (None, "bvh"):
            raise ValueError(f"unknown accel {accel!r}")
        if semantics not in ("jax", "warp"):
            raise ValueError(f"unknown semantics {semantics!r}")
        idx, t = _FirstHitFn.apply(self.vertices.contiguous(), of, df, self.triangles, self.mask, eps,
                                   0 if batch_size is None else int(batch_size),
                                   self.handle() if accel == "bvh" else None,
                                   1e-5 if semantics == "warp" else 0.0)
        return idx.reshape(batch), t.reshape(batch)
