"""Independent NumPy float32 twin of the C oracle's core arithmetic (one rounding per operation,
same association order): Moller-Trumbore, image, ray/plane, full image chain.  Bit-for-bit equality
guards the C restatement against compiler surprises (contraction, reassociation).  CPU only."""

from __future__ import annotations

import numpy as np

import oracle as orc

f = np.float32


def _dot(a, b):
    p = (a * b).astype(f)
    return ((p[..., 0] + p[..., 1]).astype(f) + p[..., 2]).astype(f)


def _cross(a, b):
    return np.stack((((a[..., 1] * b[..., 2]).astype(f) - (a[..., 2] * b[..., 1]).astype(f)).astype(f),
                     ((a[..., 2] * b[..., 0]).astype(f) - (a[..., 0] * b[..., 2]).astype(f)).astype(f),
                     ((a[..., 0] * b[..., 1]).astype(f) - (a[..., 1] * b[..., 0]).astype(f)).astype(f)), -1)


def mt_twin(o, d, tv, eps):
    """UT:1263-1322 in NumPy float32."""
    v0, v1, v2 = tv[..., 0, :], tv[..., 1, :], tv[..., 2, :]
    e1, e2 = (v1 - v0).astype(f), (v2 - v0).astype(f)
    h = _cross(d, e2)
    a = _dot(h, e1)
    a = np.where(a == 0, f(np.inf), a).astype(f)
    hit = np.abs(a) > f(eps)
    with np.errstate(all="ignore"):
        ff = (f(1.0) / a).astype(f)
        s = (o - v0).astype(f)
        u = (ff * _dot(s, h)).astype(f)
        hit &= (u >= 0) & (u <= 1)
        q = _cross(s, e1)
        v = (ff * _dot(q, d)).astype(f)
        hit &= (v >= 0) & ((u + v).astype(f) <= 1)
        t = (ff * _dot(q, e2)).astype(f)
    hit &= t > f(eps)
    return t, hit


def image_twin(x, p, n):
    c = (f(2.0) * _dot((x - p).astype(f), n)).astype(f)
    return (x - (c[..., None] * n).astype(f)).astype(f)


def ray_plane_twin(o, d, p, n):
    un, vn = _dot(d, n), _dot((p - o).astype(f), n)
    par = un == 0
    with np.errstate(all="ignore"):
        t = (vn / np.where(par, f(1.0), un)).astype(f)
    r = (o + (d * t[..., None]).astype(f)).astype(f)
    return np.where((par & (vn != 0))[..., None], f(np.inf), r).astype(f)


def test_moller_trumbore_twin(rng):
    n = 200_000
    o = rng.normal(size=(n, 3)).astype(f) * 3
    tv = (rng.normal(size=(n, 1, 3)) * 3 + rng.normal(size=(n, 3, 3))).astype(f)
    d = ((tv.mean(axis=1) + rng.normal(size=(n, 3)) * 0.7).astype(f) - o).astype(f) * f(1.7)
    for eps in (orc.DEFAULT_EPSILON, 1e-2):
        et, eh = orc.ray_intersect_triangle(o, d, tv, epsilon=eps)
        tt, th = mt_twin(o, d, tv, eps)
        np.testing.assert_array_equal(th, eh)
        np.testing.assert_array_equal(tt.view(np.uint32), et.view(np.uint32))
        assert eh.sum() > 10_000
    # exact degeneracies: rays in the triangle plane (a == 0), zero-area triangles
    o2 = np.zeros((4, 3), f)
    d2 = np.array([[1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1]], f)
    tv2 = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]] * 3 + [[[1, 1, 1], [1, 1, 1], [1, 1, 1]]], f)
    et, eh = orc.ray_intersect_triangle(o2, d2, tv2)
    tt, th = mt_twin(o2, d2, tv2, orc.DEFAULT_EPSILON)
    np.testing.assert_array_equal(th, eh)
    np.testing.assert_array_equal(np.nan_to_num(tt, nan=-7).view(np.uint32), np.nan_to_num(et, nan=-7).view(np.uint32))


def test_image_chain_twin(rng):
    B, k = 20_000, 3
    a = rng.normal(size=(B, 3)).astype(f) * 4
    b = rng.normal(size=(B, 3)).astype(f) * 4
    mv = rng.normal(size=(B, k, 3)).astype(f)
    mn, _ = orc.normalize(rng.normal(size=(B, k, 3)).astype(f))
    mn[: B // 50, 1] = np.array([0, 0, 1], f)  # some exactly axis-aligned mirrors
    np.testing.assert_array_equal(orc.image_of_vertex_with_respect_to_mirror(a, mv[:, 0], mn[:, 0]).view(np.uint32),
                                  image_twin(a, mv[:, 0], mn[:, 0]).view(np.uint32))
    np.testing.assert_array_equal(orc.intersection_of_ray_with_plane(a, (b - a).astype(f), mv[:, 0], mn[:, 0]).view(np.uint32),
                                  ray_plane_twin(a, (b - a).astype(f), mv[:, 0], mn[:, 0]).view(np.uint32))
    imgs, prev = [], a
    for j in range(k):
        prev = image_twin(prev, mv[:, j], mn[:, j])
        imgs.append(prev)
    out, cur = [None] * k, b
    for j in reversed(range(k)):
        inf = np.isinf(cur)
        pi = np.where(inf, f(0), cur).astype(f)
        x = ray_plane_twin(pi, (imgs[j] - pi).astype(f), mv[:, j], mn[:, j])
        cur = np.where(inf, f(np.inf), x).astype(f)
        out[j] = cur
    twin = np.stack(out, axis=1)
    np.testing.assert_array_equal(orc.image_method(a, b, mv, mn).view(np.uint32), twin.view(np.uint32))
