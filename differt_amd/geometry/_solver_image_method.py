"""Image-method operators with the reference's signatures.

Mirrors ``differt/src/differt/geometry/_solver_image_method.py``; arithmetic in HIP
(``csrc/image_method.hip``, ``csrc/image_chain.hpp``).  ``image_method`` is differentiable
(``torch.autograd``) through a hand-derived VJP kernel.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._tensors import as_f32, device, ptr, stream

__all__ = [
    "consecutive_vertices_are_on_same_side_of_mirror",
    "image_method",
    "image_of_vertex_with_respect_to_mirror",
    "intersection_of_ray_with_plane",
]


def _bcast(batch, *arrs_and_tails):
    out = []
    for a, tail in arrs_and_tails:
        out.append(a.expand(*batch, *tail).contiguous())
    return out


def _rows(x: torch.Tensor, dense: int):
    """``(tensor whose pointer to pass, stride in floats)`` of a ``[B, ...]`` input: a row shared by the whole batch
    (an expanded view, stride 0) is read in place by the kernel instead of being materialised ``B`` times."""
    if x.shape[0] > 1 and x.stride(0) == 0 and x[0].is_contiguous():
        return x[0], 0
    return x.contiguous(), dense


class _ImageMethodFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mv, mn):
        B, k = mv.shape[0], mv.shape[1]
        out = torch.empty((B, k, 3), dtype=torch.float32, device=mv.device)
        (pa, sa), (pb, sb), (pmv, smv), (pmn, smn) = _rows(a, 3), _rows(b, 3), _rows(mv, 3 * k), _rows(mn, 3 * k)
        _lib.call("drt_image_method_strided", ptr(pa), sa, ptr(pb), sb, ptr(pmv), smv, ptr(pmn), smn, B, k, ptr(out),
                  stream())
        ctx.save_for_backward(a, b, mv, mn)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, mv, mn = ctx.saved_tensors
        B, k = mv.shape[0], mv.shape[1]
        g = g.contiguous()
        dev = g.device
        # per-element gradients; autograd sums those of rows that were broadcast (expand's backward)
        ga, gb = (torch.empty((B, 3), dtype=torch.float32, device=dev) for _ in range(2))
        gmv, gmn = (torch.empty((B, k, 3), dtype=torch.float32, device=dev) for _ in range(2))
        (pa, sa), (pb, sb), (pmv, smv), (pmn, smn) = _rows(a, 3), _rows(b, 3), _rows(mv, 3 * k), _rows(mn, 3 * k)
        _lib.call("drt_image_method_vjp_strided", ptr(pa), sa, ptr(pb), sb, ptr(pmv), smv, ptr(pmn), smn, ptr(g), B, k,
                  ptr(ga), ptr(gb), ptr(gmv), ptr(gmn), stream())
        return ga, gb, gmv, gmn


def _flat_rows(x: torch.Tensor, batch, B: int, tail) -> torch.Tensor:
    """``x`` broadcast to ``[*batch, *tail]`` as a ``[B, *tail]`` tensor: a stride-0 view when ``x`` is one row."""
    if x.numel() == int(np.prod(tail, dtype=np.int64)):
        return x.reshape(1, *tail).expand(B, *tail)
    return x.expand(*batch, *tail).reshape(B, *tail)


def image_method(from_vertex, to_vertex, mirror_vertices, mirror_normals):
    """Path vertices (end points excluded) reflecting on the given mirrors, ``[*batch, k, 3]``.

    Reference: ``image_method`` _solver_image_method.py:206-363 (forward scan of images :191-195,
    reverse scan of ray/plane intersections :196-201, inf propagation :165-181, ``k == 0`` -> empty
    :349-358; broadcasting ``(3),(3),(n,3),(n,3)->(n,3)`` :360-363)."""
    dev = device()
    a, b = as_f32(from_vertex, dev), as_f32(to_vertex, dev)
    mv, mn = as_f32(mirror_vertices, dev), as_f32(mirror_normals, dev)
    k = mv.shape[-2]
    batch = torch.broadcast_shapes(a.shape[:-1], b.shape[:-1], mv.shape[:-2], mn.shape[:-2])
    B = int(np.prod(batch, dtype=np.int64))
    if k == 0 or B == 0:
        return torch.empty((*batch, k, 3), dtype=torch.float32, device=dev)
    out = _ImageMethodFn.apply(_flat_rows(a, batch, B, (3,)), _flat_rows(b, batch, B, (3,)),
                               _flat_rows(mv, batch, B, (k, 3)), _flat_rows(mn, batch, B, (k, 3)))
    return out.reshape(*batch, k, 3)


def image_of_vertex_with_respect_to_mirror(vertex, mirror_vertex, mirror_normal):
    """``x - 2<x-p,n>n`` (_solver_image_method.py:11-79)."""
    dev = device()
    x, p, n = as_f32(vertex, dev), as_f32(mirror_vertex, dev), as_f32(mirror_normal, dev)
    batch = torch.broadcast_shapes(x.shape[:-1], p.shape[:-1], n.shape[:-1])
    B = int(np.prod(batch, dtype=np.int64))
    out = torch.empty((*batch, 3), dtype=torch.float32, device=dev)
    if B:
        x, p, n = _bcast(batch, (x, (3,)), (p, (3,)), (n, (3,)))
        _lib.call("drt_image_of_vertex", ptr(x), ptr(p), ptr(n), B, ptr(out), stream())
    return out


def intersection_of_ray_with_plane(ray_origin, ray_direction, plane_vertex, plane_normal):
    """_solver_image_method.py:82-135 (parallel rays -> inf, or the origin if it is on the plane)."""
    dev = device()
    o, d = as_f32(ray_origin, dev), as_f32(ray_direction, dev)
    p, n = as_f32(plane_vertex, dev), as_f32(plane_normal, dev)
    batch = torch.broadcast_shapes(o.shape[:-1], d.shape[:-1], p.shape[:-1], n.shape[:-1])
    B = int(np.prod(batch, dtype=np.int64))
    out = torch.empty((*batch, 3), dtype=torch.float32, device=dev)
    if B:
        o, d, p, n = _bcast(batch, (o, (3,)), (d, (3,)), (p, (3,)), (n, (3,)))
        _lib.call("drt_intersection_of_ray_with_plane", ptr(o), ptr(d), ptr(p), ptr(n), B, ptr(out),
                  stream())
    return out


def consecutive_vertices_are_on_same_side_of_mirror(
    vertices, mirror_vertices, mirror_normals, *, smoothing_factor=None
):
    """_solver_image_method.py:386-454; needs ``num_vertices == num_mirrors + 2``.  With
    ``smoothing_factor`` the result is ``sigmoid(alpha * sign(dot_prev) * sign(dot_next))`` (:450-453),
    a float whose gradient is zero everywhere (``sign`` is piecewise constant)."""
    dev = device()
    v, mv, mn = as_f32(vertices, dev), as_f32(mirror_vertices, dev), as_f32(mirror_normals, dev)
    k = mv.shape[-2]
    if v.shape[-2] != k + 2:  # chex.assert_axis_dimension(..., exception_type=TypeError), :422-424
        raise TypeError(f"expected vertices with {k + 2} points on axis -2, got {v.shape[-2]}")
    batch = torch.broadcast_shapes(v.shape[:-2], mv.shape[:-2], mn.shape[:-2])
    B = int(np.prod(batch, dtype=np.int64))
    if smoothing_factor is not None:
        out = torch.empty((*batch, k), dtype=torch.float32, device=dev)
        if k and B:
            v, mv, mn = _bcast(batch, (v.detach(), (k + 2, 3)), (mv.detach(), (k, 3)), (mn.detach(), (k, 3)))
            _lib.call("drt_consecutive_vertices_same_side_smooth", ptr(v), ptr(mv), ptr(mn), B, k,
                      float(smoothing_factor), ptr(out), stream())
        return out
    out = torch.empty((*batch, k), dtype=torch.uint8, device=dev)
    if k and B:
        v, mv, mn = _bcast(batch, (v, (k + 2, 3)), (mv, (k, 3)), (mn, (k, 3)))
        _lib.call("drt_consecutive_vertices_same_side", ptr(v), ptr(mv), ptr(mn), B, k, ptr(out), stream())
    return out.bool()
