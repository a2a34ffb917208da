"""Ray/triangle operators and path-candidate wrappers with the reference's signatures.

Mirrors ``differt/src/differt/geometry/_utils.py`` of the reference (file:line in each docstring);
the arithmetic runs in hand-written HIP kernels behind the C ABI (``include/differt_amd.h``).
Inputs may be NumPy arrays, sequences or torch tensors; outputs are torch tensors in HBM.
"""

from __future__ import annotations

from collections.abc import Iterator
from typing import Any

import numpy as np
import torch

from .. import _lib
from .._tensors import F32_EPS, as_f32, as_u8, device, ptr, stream

__all__ = [
    "SizedIterator",
    "assemble_path",
    "cartesian_to_spherical",
    "fibonacci_lattice",
    "spherical_to_cartesian",
    "triangles_visible_from_vertex",
    "viewing_frustum",
    "first_triangle_hit_by_ray",
    "generate_all_path_candidates",
    "generate_all_path_candidates_chunks_iter",
    "generate_all_path_candidates_iter",
    "normalize",
    "ray_intersect_any_triangle",
    "ray_intersect_triangle",
]


def _no_smoothing(smoothing_factor) -> None:
    if smoothing_factor is not None:
        raise NotImplementedError(
            "smoothing_factor is not defined for this operator (the reference only smooths "
            "ray_intersect_triangle / ray_intersect_any_triangle / the tracer)"
        )


class _MtSmoothFn(torch.autograd.Function):
    """Smoothed Moller-Trumbore (_utils.py:1279-1320) on flat buffers: rays ``[R,3]``, triangles
    ``[T,3,3]``; ``dense`` -> ``[R,T]`` outputs, else paired ``[R]``.  Differentiable in all three."""

    @staticmethod
    def forward(ctx, o, d, tv, eps, alpha, dense):
        R, T = o.shape[0], tv.shape[0]
        shape = (R, T) if dense else (R,)
        t = torch.empty(shape, dtype=torch.float32, device=o.device)
        hit = torch.empty(shape, dtype=torch.float32, device=o.device)
        if t.numel():
            _lib.call("drt_ray_intersect_triangle_smooth", ptr(o), ptr(d), R, ptr(tv), T, int(dense), eps,
                      alpha, ptr(t), ptr(hit), stream())
        ctx.save_for_backward(o, d, tv)
        ctx.cfg = (eps, alpha, dense)
        return t, hit

    @staticmethod
    def backward(ctx, gt, gh):
        o, d, tv = ctx.saved_tensors
        eps, alpha, dense = ctx.cfg
        go, gd, gtv = torch.zeros_like(o), torch.zeros_like(d), torch.zeros_like(tv)
        if o.numel() and tv.numel():
            _lib.call("drt_ray_intersect_triangle_smooth_vjp", ptr(o), ptr(d), o.shape[0], ptr(tv),
                      tv.shape[0], int(dense), eps, alpha,
                      ptr(None if gt is None else gt.contiguous()),
                      ptr(None if gh is None else gh.contiguous()), ptr(go), ptr(gd), ptr(gtv), stream())
        return go, gd, gtv, None, None, None


class _MtHardFn(torch.autograd.Function):
    """Hard-mode Moller-Trumbore on flat buffers with the reference's differentiable ``t``
    (_utils.py:1316: plain JAX arithmetic, so ``jax.grad`` flows through ``t``; ``hit`` is a bool).
    ``dense``: rays ``[R,3]`` x triangles ``[T,3,3]`` -> ``[R,T]``; else paired ``[R]``."""

    @staticmethod
    def forward(ctx, o, d, tv, eps, dense):
        R, T = o.shape[0], tv.shape[0]
        shape = (R, T) if dense else (R,)
        t = torch.empty(shape, dtype=torch.float32, device=o.device)
        hit = torch.empty(shape, dtype=torch.uint8, device=o.device)
        if t.numel():
            if dense:
                _lib.call("drt_ray_intersect_triangle_dense", ptr(o), ptr(d), R, ptr(tv), T, eps, ptr(t), ptr(hit),
                          stream())
            else:
                _lib.call("drt_ray_intersect_triangle_paired", ptr(o), ptr(d), ptr(tv), R, eps, ptr(t), ptr(hit),
                          stream())
        ctx.save_for_backward(o, d, tv)
        ctx.dense = dense
        ctx.mark_non_differentiable(hit)
        return t, hit

    @staticmethod
    def backward(ctx, gt, _gh):
        o, d, tv = ctx.saved_tensors
        need = ctx.needs_input_grad
        go = torch.zeros_like(o) if need[0] else None
        gd = torch.zeros_like(d) if need[1] else None
        gtv = torch.zeros_like(tv) if need[2] else None
        if o.numel() and tv.numel() and gt is not None:
            _lib.call("drt_ray_intersect_triangle_vjp", ptr(o), ptr(d), o.shape[0], ptr(tv), tv.shape[0],
                      int(ctx.dense), ptr(gt.contiguous()), ptr(go), ptr(gd), ptr(gtv), stream())
        return go, gd, gtv, None, None


class _AnySmoothFn(torch.autograd.Function):
    """Smoothed any-triangle confidence (_utils.py:1436-1537) on the flat layout of
    :func:`_flatten_query`; differentiable in the rays and the triangle vertices."""

    @staticmethod
    def forward(ctx, o, d, tv, act, cfg):
        R, T, tvs, acts, eps, tol, alpha, bs = cfg
        out = torch.zeros(R, dtype=torch.float32, device=o.device)
        if R:
            _lib.call("drt_ray_intersect_any_triangle_smooth", ptr(o), ptr(d), R, ptr(tv), T, tvs, ptr(act),
                      acts, eps, tol, alpha, bs, ptr(out), stream())
        ctx.save_for_backward(o, d, tv)
        ctx.act, ctx.cfg = act, cfg
        return out

    @staticmethod
    def backward(ctx, gout):
        o, d, tv = ctx.saved_tensors
        R, T, tvs, acts, eps, tol, alpha, bs = ctx.cfg
        go, gd, gtv = torch.zeros_like(o), torch.zeros_like(d), torch.zeros_like(tv)
        if R and T:
            _lib.call("drt_ray_intersect_any_triangle_smooth_vjp", ptr(o), ptr(d), R, ptr(tv), T, tvs,
                      ptr(ctx.act), acts, eps, tol, alpha, bs, ptr(gout.contiguous()), ptr(go), ptr(gd),
                      ptr(gtv), stream())
        return go, gd, gtv, None, None


def normalize(vectors, keepdims: bool = False):
    """Reference ``normalize`` (_utils.py:29-72): zero-length vectors are divided by one.
    Returns ``(unit vectors, lengths)``."""
    dev = device()
    v = as_f32(vectors, dev).contiguous()
    batch = v.shape[:-1]
    B = int(np.prod(batch, dtype=np.int64))
    out = torch.empty_like(v)
    lengths = torch.empty(batch, dtype=torch.float32, device=dev)
    if B:
        _lib.call("drt_normalize", ptr(v.reshape(B, 3)), B, ptr(out), ptr(lengths), stream())
    return out, (lengths[..., None] if keepdims else lengths)


def assemble_path(from_vertex, intermediate_vertices, to_vertex=None):
    """Reference ``assemble_path`` (_utils.py:514-565): concatenate [from, inter..., to]."""
    a = as_f32(from_vertex)
    m = as_f32(intermediate_vertices)
    if to_vertex is None:
        b = m
        batch = torch.broadcast_shapes(a.shape[:-1], b.shape[:-1])
        return torch.cat(
            (a[..., None, :].expand(*batch, 1, 3), b[..., None, :].expand(*batch, 1, 3)), dim=-2
        )
    b = as_f32(to_vertex)
    batch = torch.broadcast_shapes(a.shape[:-1], m.shape[:-2], b.shape[:-1])
    return torch.cat(
        (
            a[..., None, :].expand(*batch, 1, 3),
            m.expand(*batch, *m.shape[-2:]),
            b[..., None, :].expand(*batch, 1, 3),
        ),
        dim=-2,
    )


def ray_intersect_triangle(
    ray_origins,
    ray_directions,
    triangle_vertices,
    *,
    epsilon: float | None = None,
    smoothing_factor=None,
):
    """Moller-Trumbore for every broadcast (ray, triangle) pair; returns ``(t, hit)``.

    Reference: ``ray_intersect_triangle`` _utils.py:1157-1322 (default ``epsilon = 10*eps``,
    :1257-1259; ``t`` is returned for misses too).  The ``o[..., None, :]`` x ``tv[T,3,3]`` outer
    form runs the dense kernel (no input materialisation); any other broadcast runs paired.
    """
    dev = device()
    o, d, tv = as_f32(ray_origins, dev), as_f32(ray_directions, dev), as_f32(triangle_vertices, dev)
    eps = 10.0 * F32_EPS if epsilon is None else float(epsilon)
    batch = torch.broadcast_shapes(o.shape[:-1], d.shape[:-1], tv.shape[:-2])
    sf = None if smoothing_factor is None else float(smoothing_factor)
    # outer-product form: rays [..., 1, 3] against one shared triangle list [T, 3, 3]
    nb = len(batch)
    tvb = (1,) * (nb - (tv.dim() - 2)) + tuple(tv.shape[:-2])
    ob = (1,) * (nb - (o.dim() - 1)) + tuple(o.shape[:-1])
    db = (1,) * (nb - (d.dim() - 1)) + tuple(d.shape[:-1])
    dense = (
        nb >= 1
        and all(s == 1 for s in tvb[:-1])
        and ob[-1] == 1
        and db[-1] == 1
    )
    if sf is not None:  # _utils.py:1279-1320: hit is a float confidence; differentiable (autograd)
        if dense:
            T, rb = batch[-1], batch[:-1]
            of = o.reshape(ob[:-1] + (3,)).expand(*rb, 3).contiguous().reshape(-1, 3)
            df = d.reshape(db[:-1] + (3,)).expand(*rb, 3).contiguous().reshape(-1, 3)
            t, hit = _MtSmoothFn.apply(of, df, tv.reshape(T, 3, 3).contiguous(), eps, sf, True)
        else:
            n = int(np.prod(batch, dtype=np.int64))
            of = o.expand(*batch, 3).contiguous().reshape(n, 3)
            df = d.expand(*batch, 3).contiguous().reshape(n, 3)
            tvf = tv.expand(*batch, 3, 3).contiguous().reshape(n, 3, 3)
            t, hit = _MtSmoothFn.apply(of, df, tvf, eps, sf, False)
        return t.reshape(batch), hit.reshape(batch)
    if torch.is_grad_enabled() and (o.requires_grad or d.requires_grad or tv.requires_grad):
        # differentiable `t` like the reference (the flat views below are autograd-tracked, so the
        # broadcasting is undone by torch: expanded inputs receive summed gradients)
        if dense:
            T, rb = batch[-1], batch[:-1]
            of = o.reshape(ob[:-1] + (3,)).expand(*rb, 3).contiguous().reshape(-1, 3)
            df = d.reshape(db[:-1] + (3,)).expand(*rb, 3).contiguous().reshape(-1, 3)
            t, hit = _MtHardFn.apply(of, df, tv.reshape(T, 3, 3).contiguous(), eps, True)
        else:
            n = int(np.prod(batch, dtype=np.int64))
            of = o.expand(*batch, 3).contiguous().reshape(n, 3)
            df = d.expand(*batch, 3).contiguous().reshape(n, 3)
            tvf = tv.expand(*batch, 3, 3).contiguous().reshape(n, 3, 3)
            t, hit = _MtHardFn.apply(of, df, tvf, eps, False)
        return t.reshape(batch), hit.reshape(batch).bool()
    t = torch.empty(batch, dtype=torch.float32, device=dev)
    hit = torch.empty(batch, dtype=torch.uint8, device=dev)
    if t.numel() == 0:
        return t, hit.bool()
    if dense:
        T = batch[-1]
        rb = batch[:-1]
        of = o.reshape(ob[:-1] + (3,)).expand(*rb, 3).contiguous().reshape(-1, 3)
        df = d.reshape(db[:-1] + (3,)).expand(*rb, 3).contiguous().reshape(-1, 3)
        tvf = tv.reshape(T, 3, 3).contiguous()
        _lib.call(
            "drt_ray_intersect_triangle_dense",
            ptr(of), ptr(df), of.shape[0], ptr(tvf), T, eps, ptr(t), ptr(hit), stream(),
        )
    elif nb >= 3 and ob[-1] == 1 and db[-1] == 1 and tvb[-2] == 1 and int(np.prod(batch[:-2], dtype=np.int64)) <= 65535:
        # the outer form under leading batch axes (`vmap` of configs[1]: o [*L,R,1,3] x tv [*L,1,T,3,3],
        # docs/source/batch_axes.md:67-80): B = prod(L) independent R x T problems in ONE launch -- a single
        # 256-ray problem is latency-bound, a batch of them is not
        lead, R, T = batch[:-2], batch[-2], batch[-1]
        B = int(np.prod(lead, dtype=np.int64))
        of = o.reshape(ob[:-1] + (3,)).expand(*lead, R, 3).contiguous()
        df = d.reshape(db[:-1] + (3,)).expand(*lead, R, 3).contiguous()
        shared_tv = all(x == 1 for x in tvb[:-2])
        tvf = (tv.reshape(T, 3, 3) if shared_tv
               else tv.reshape(tvb[:-2] + (T, 3, 3)).expand(*lead, T, 3, 3)).contiguous()
        _lib.call(
            "drt_ray_intersect_triangle_dense_batched",
            ptr(of), ptr(df), 3 * R, R, ptr(tvf), 0 if shared_tv else 9 * T, T, B, eps, ptr(t), ptr(hit), stream(),
        )
    else:
        of = o.expand(*batch, 3).contiguous()
        df = d.expand(*batch, 3).contiguous()
        tvf = tv.expand(*batch, 3, 3).contiguous()
        _lib.call(
            "drt_ray_intersect_triangle_paired",
            ptr(of), ptr(df), ptr(tvf), t.numel(), eps, ptr(t), ptr(hit), stream(),
        )
    return t, hit.bool()


def _flatten_query(ray_origins, ray_directions, triangle_vertices, active_triangles):
    dev = device()
    o, d, tv = as_f32(ray_origins, dev), as_f32(ray_directions, dev), as_f32(triangle_vertices, dev)
    T = tv.shape[-3]
    act = None if active_triangles is None else as_u8(active_triangles, dev)
    batch = torch.broadcast_shapes(
        o.shape[:-1], d.shape[:-1], tv.shape[:-3], act.shape[:-1] if act is not None else ()
    )
    R = int(np.prod(batch, dtype=np.int64))
    of = o.expand(*batch, 3).contiguous().reshape(R, 3)
    df = d.expand(*batch, 3).contiguous().reshape(R, 3)
    if all(s == 1 for s in tv.shape[:-3]):
        tvf, tv_stride = tv.reshape(T, 3, 3).contiguous(), 0
    else:
        tvf, tv_stride = tv.expand(*batch, T, 3, 3).contiguous().reshape(R, T, 3, 3), 9 * T
    if act is None:
        actf, act_stride = None, 0
    elif all(s == 1 for s in act.shape[:-1]):
        actf, act_stride = act.reshape(T).contiguous(), 0
    else:
        actf, act_stride = act.expand(*batch, T).contiguous().reshape(R, T), T
    return dev, batch, R, T, of, df, tvf, tv_stride, actf, act_stride


def ray_intersect_any_triangle(
    ray_origins,
    ray_directions,
    triangle_vertices,
    active_triangles=None,
    *,
    hit_tol: float | None = None,
    smoothing_factor=None,
    batch_size: int | None = 512,  # tiling does not change an OR; it does order the smoothed sums
    **kwargs: Any,
):
    """Whether each ray hits any triangle before ``t = 1 - hit_tol``.

    Reference: ``ray_intersect_any_triangle`` _utils.py:1353-1537 (``hit_tol = 100*eps`` default
    :1418-1420, ``T == 0 -> False`` :1441-1450, ``active_triangles`` :1468-1469).  ``epsilon`` is
    forwarded to Moller-Trumbore through ``**kwargs`` like in the reference.
    """
    epsilon = kwargs.pop("epsilon", None)
    if kwargs:
        raise TypeError(f"unexpected keyword arguments: {sorted(kwargs)}")
    dev, batch, R, T, o, d, tv, tvs, act, acts = _flatten_query(
        ray_origins, ray_directions, triangle_vertices, active_triangles
    )
    eps = 10.0 * F32_EPS if epsilon is None else float(epsilon)
    tol = 100.0 * F32_EPS if hit_tol is None else float(hit_tol)
    if smoothing_factor is not None:  # float confidence, :1465-1476 (tiles matter: clipped sums)
        cfg = (R, T, tvs, acts, eps, tol, float(smoothing_factor), 0 if batch_size is None else int(batch_size))
        return _AnySmoothFn.apply(o, d, tv, act, cfg).reshape(batch)
    out = torch.zeros(R, dtype=torch.uint8, device=dev)
    if R:
        _lib.call(
            "drt_ray_intersect_any_triangle",
            ptr(o), ptr(d), R, ptr(tv), T, tvs, ptr(act), acts, eps, tol, ptr(out), stream(),
        )
    return out.bool().reshape(batch)


def first_triangle_hit_by_ray(
    ray_origins,
    ray_directions,
    triangle_vertices,
    active_triangles=None,
    batch_size: int | None = 512,
    **kwargs: Any,
):
    """Index of and distance to the first triangle hit by each ray; miss = ``(-1, inf)``.

    Reference: ``first_triangle_hit_by_ray`` _utils.py:1775-1960, including the tie-break that its
    tiling implies (lowest index inside a ``batch_size`` tile, the later tile wins: :1865-1867,
    :1886; remainder tile last :1939-1955), so hit indices are bit-exact with the reference.
    """
    epsilon = kwargs.pop("epsilon", None)
    _no_smoothing(kwargs.pop("smoothing_factor", None))
    if kwargs:
        raise TypeError(f"unexpected keyword arguments: {sorted(kwargs)}")
    dev, batch, R, T, o, d, tv, tvs, act, acts = _flatten_query(
        ray_origins, ray_directions, triangle_vertices, active_triangles
    )
    eps = 10.0 * F32_EPS if epsilon is None else float(epsilon)
    idx = torch.full((R,), -1, dtype=torch.int32, device=dev)
    t = torch.full((R,), float("inf"), dtype=torch.float32, device=dev)
    if R:
        ws = torch.empty(R, dtype=torch.int64, device=dev)
        _lib.call(
            "drt_first_triangle_hit_by_ray",
            ptr(o), ptr(d), R, ptr(tv), T, tvs, ptr(act), acts, eps,
            0 if batch_size is None else int(batch_size), ptr(idx), ptr(t), ptr(ws), R * 8, stream(),
        )
    if R and T and torch.is_grad_enabled() and (o.requires_grad or d.requires_grad or tv.requires_grad):
        # the reference's t is differentiable (argmin picks one triangle, t = min over the hits,
        # _utils.py:1886-1960): re-evaluate the paired operator on the hit triangle -- the same
        # arithmetic, hence the same bits -- with its VJP attached; misses keep the constant inf
        hitm = idx >= 0
        sel = idx.clamp(min=0).long()
        rows = torch.arange(R, device=dev)
        tv_hit = tv[sel] if tvs == 0 else tv[rows, sel]
        t_hit, _ = _MtHardFn.apply(o, d, tv_hit.contiguous(), eps, False)
        t = torch.where(hitm, t_hit, t)
    return idx.reshape(batch), t.reshape(batch)


# ------------------------------------------------------------------------------------------
# visibility by ray launching (reference _utils.py:369-490, 639-993, 1540-1772)
# ------------------------------------------------------------------------------------------
def cartesian_to_spherical(xyz):
    """Reference ``cartesian_to_spherical`` (_utils.py:930-958): ``(r, polar, azimuth)``."""
    x = as_f32(xyz).detach().contiguous()
    out = torch.empty_like(x)
    B = x.numel() // 3
    if B:
        _lib.call("drt_cartesian_to_spherical", ptr(x), B, ptr(out), stream())
    return out


def spherical_to_cartesian(rpa):
    """Reference ``spherical_to_cartesian`` (_utils.py:961-993); radius 1 when missing."""
    v = as_f32(rpa).detach().contiguous()
    w = v.shape[-1]
    out = torch.empty((*v.shape[:-1], 3), dtype=torch.float32, device=v.device)
    B = v.numel() // w if w else 0
    if B:
        _lib.call("drt_spherical_to_cartesian", ptr(v), B, w, ptr(out), stream())
    return out


def fibonacci_lattice(n: int, dtype=None, *, frustum=None):  # noqa: ARG001
    """``n`` directions on the unit sphere, or inside ``frustum [2, 2|3]`` (_utils.py:369-490,
    including the split-modulus evaluation of ``frac(i / phi)`` for large ``n``, :426-462)."""
    if n <= 0:
        raise ValueError(f"Invalid size {n!r}, must be strictly positive.")
    dev = device()
    out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    fr = None
    if frustum is not None:
        f = as_f32(frustum, dev)
        fr = torch.zeros((2, 3), dtype=torch.float32, device=dev)
        fr[:, 3 - f.shape[-1]:] = f.reshape(2, -1)
    _lib.call("drt_fibonacci_lattice", n, ptr(fr), ptr(out), stream())
    return out


def viewing_frustum(viewing_vertex, world_vertices, *, active_vertices=None, reduce: bool = False):
    """Spherical bounding region of the world seen from a vertex, reference signature and semantics
    (_utils.py:639-927): ``viewing_vertex [*#batch, 3]``, ``world_vertices [*#batch, N, 3]`` POINTS,
    ``active_vertices [*#batch, N]``; returns ``[*batch, 2, 3]`` (r, polar, azimuth extents), or one
    ``[2, 3]`` frustum over every batch entry with ``reduce=True``."""
    dev = device()
    v = as_f32(viewing_vertex, dev)
    w = as_f32(world_vertices, dev)
    act = None if active_vertices is None else as_u8(active_vertices, dev)
    N = w.shape[-2]
    batch = torch.broadcast_shapes(v.shape[:-1], w.shape[:-2], act.shape[:-1] if act is not None else ())
    B = int(np.prod(batch, dtype=np.int64))
    vf = v.expand(*batch, 3).contiguous().reshape(B, 3)
    if all(s_ == 1 for s_ in w.shape[:-2]):
        wf, wstride = w.reshape(N, 3).contiguous(), 0
    else:
        wf, wstride = w.expand(*batch, N, 3).contiguous().reshape(B, N, 3), 3 * N
    if act is None:
        af, astride = None, 0
    elif all(s_ == 1 for s_ in act.shape[:-1]):
        af, astride = act.reshape(N).contiguous(), 0
    else:
        af, astride = act.expand(*batch, N).contiguous().reshape(B, N), N
    out = torch.empty((1 if reduce else max(B, 1), 2, 3), dtype=torch.float32, device=dev)
    ws = torch.empty((max(B, 1), 8), dtype=torch.float32, device=dev) if reduce else None
    if B:
        _lib.call("drt_viewing_frustum_general", ptr(vf), B, ptr(wf), N, wstride, ptr(af), astride, int(reduce),
                  ptr(ws), ptr(out), stream())
    return out.reshape(2, 3) if reduce else out[:B].reshape(*batch, 2, 3)


def triangles_visible_from_vertex(
    vertex,
    triangle_vertices,
    active_triangles=None,
    num_rays: int = int(1e6),
    batch_size: int | None = 512,  # noqa: ARG001 - rays are generated on the fly, nothing to batch
    **kwargs: Any,
):
    """Which triangles receive the first hit of at least one of ``num_rays`` lattice rays launched
    inside the viewing frustum (_utils.py:1540-1772).  ``bool[*batch, T]``; ``triangle_vertices`` /
    ``active_triangles`` may carry their own broadcastable batch (``*#batch``) like in the reference:
    every distinct triangle set gets its own launch."""
    epsilon = kwargs.pop("epsilon", None)
    if kwargs:
        raise TypeError(f"unexpected keyword arguments: {sorted(kwargs)}")
    dev = device()
    tv = as_f32(triangle_vertices, dev)
    act = None if active_triangles is None else as_u8(active_triangles, dev)
    v = as_f32(vertex, dev)
    T = tv.shape[-3]
    eps = 10.0 * F32_EPS if epsilon is None else float(epsilon)

    def launch(vf, tvf, actf):
        B = vf.shape[0]
        vis = torch.zeros((B, T), dtype=torch.uint8, device=dev)
        if B and T:
            ws = torch.empty((B, 6), dtype=torch.float32, device=dev)
            _lib.call("drt_triangles_visible_from_vertex", ptr(vf), B, ptr(tvf), T, ptr(actf), int(num_rays),
                      eps, ptr(vis), ptr(ws), stream())
        return vis

    shared_tv = all(s_ == 1 for s_ in tv.shape[:-3])
    shared_act = act is None or all(s_ == 1 for s_ in act.shape[:-1])
    if shared_tv and shared_act:
        batch = v.shape[:-1]
        vis = launch(v.reshape(-1, 3).contiguous(), tv.reshape(T, 3, 3).contiguous(),
                     None if act is None else act.reshape(T).contiguous())
        return vis.bool().reshape(*batch, T)
    batch = torch.broadcast_shapes(v.shape[:-1], tv.shape[:-3], act.shape[:-1] if act is not None else ())
    B = int(np.prod(batch, dtype=np.int64))
    vf = v.expand(*batch, 3).reshape(B, 3)
    tvb = tv.expand(*batch, T, 3, 3).reshape(B, T, 3, 3)
    actb = None if act is None else act.expand(*batch, T).reshape(B, T)
    rows = [launch(vf[b:b + 1].contiguous(), tvb[b].contiguous(), None if actb is None else actb[b].contiguous())
            for b in range(B)]
    return (torch.cat(rows) if rows else torch.zeros((0, T), dtype=torch.uint8, device=dev)).bool().reshape(*batch, T)


# ------------------------------------------------------------------------------------------
# path candidates (reference _utils.py:1004-1132 over differt-core graph.rs)
# ------------------------------------------------------------------------------------------
class SizedIterator:
    """Reference ``SizedIterator`` (_utils.py:1004-1044): an iterator with a known length."""

    def __init__(self, iter_: Iterator, size) -> None:
        self.iter_ = iter_
        self.size = size

    def __iter__(self):
        return self

    def __next__(self):
        return next(self.iter_)

    def __len__(self) -> int:
        return self.size() if callable(self.size) else self.size


def generate_all_path_candidates(num_primitives: int, order: int) -> np.ndarray:
    """All path candidates, lexicographic, shape ``[n*(n-1)**(order-1), order]`` (int32, host).

    Reference: ``generate_all_path_candidates`` _utils.py:1047-1081 over
    ``CompleteGraph.all_paths_array`` (graph.rs:222-233).  The table is produced by closed-form
    unranking in the native library (no odometer).
    """
    from ._graph import CompleteGraph

    return CompleteGraph(num_primitives).all_paths_array(
        num_primitives, num_primitives + 1, order + 2, include_from_and_to=False
    ).astype(np.int32)


def generate_all_path_candidates_iter(num_primitives: int, order: int) -> SizedIterator:
    """Reference ``generate_all_path_candidates_iter`` (_utils.py:1084-1105)."""
    from ._graph import CompleteGraph

    it = CompleteGraph(num_primitives).all_paths(
        num_primitives, num_primitives + 1, order + 2, include_from_and_to=False
    )
    return SizedIterator((np.asarray(a, dtype=np.int32) for a in it), size=it.__len__)


def generate_all_path_candidates_chunks_iter(
    num_primitives: int, order: int, chunk_size: int = 1000
) -> SizedIterator:
    """Reference ``generate_all_path_candidates_chunks_iter`` (_utils.py:1108-1132)."""
    from ._graph import CompleteGraph

    it = CompleteGraph(num_primitives).all_paths_array_chunks(
        num_primitives, num_primitives + 1, order + 2, include_from_and_to=False, chunk_size=chunk_size
    )
    return SizedIterator((np.asarray(a, dtype=np.int32) for a in it), size=it.__len__)
