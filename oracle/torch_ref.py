"""Torch (CPU, float64/float32) restatement of the DIFFERENTIABLE part of the hot path.

TEST INFRASTRUCTURE ONLY -- the gradient oracle.  The reference differentiates these functions with
JAX autodiff; no reference test pins the image-method gradients numerically,
so parity of the hand-written VJP kernels is checked against ``torch.autograd`` over this literal
restatement (and against float64 central differences in the tests).

Follows /root/reference/differt/src/differt/geometry/_solver_image_method.py:68-79, 110-135,
138-203 including the where-guards that keep gradients NaN-free, _mesh.py:226-255
(`_differentiable_distance`) and _mesh.py:950-956 (normals).
"""

from __future__ import annotations

import torch


def image_of_vertex(x, p, n):
    inc = x - p
    return x - 2.0 * (inc * n).sum(-1, keepdim=True) * n  # IM:73-79


def intersection_of_ray_with_plane(o, d, p, n):
    v = p - o
    un = (d * n).sum(-1, keepdim=True)
    vn = (v * n).sum(-1, keepdim=True)
    parallel = un == 0.0
    un = torch.where(parallel, torch.ones_like(un), un)  # IM:123-124
    t = vn / un
    res = o + d * t
    return torch.where(parallel & (vn != 0.0), torch.full_like(res, float("inf")), res)  # IM:131-135


def image_method(a, b, mv, mn):
    """a, b: [..., 3]; mv, mn: [..., k, 3] -> [..., k, 3] (IM:185-203)."""
    k = mv.shape[-2]
    images, prev = [], a
    for j in range(k):
        prev = image_of_vertex(prev, mv[..., j, :], mn[..., j, :])
        images.append(prev)
    out, cur = [None] * k, b
    for j in reversed(range(k)):
        noprev = torch.isinf(cur)  # IM:165-170
        pi = torch.where(noprev, torch.zeros_like(cur), cur)
        x = intersection_of_ray_with_plane(pi, images[j] - pi, mv[..., j, :], mn[..., j, :])
        cur = torch.where(noprev, torch.full_like(x, float("inf")), x)  # IM:177-181
        out[j] = cur
    return torch.stack(out, dim=-2)


def normals(vertices, triangles):
    tv = vertices[triangles]
    c = torch.linalg.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 1])  # ME:950-956
    ln = torch.sqrt((c * c).sum(-1, keepdim=True))
    return c / torch.where(ln == 0.0, torch.ones_like(ln), ln)


def trace_vertices(vertices, triangles, tx, rx, cand):
    """Full path vertices [Ntx, Nrx, C, k+2, 3] of SV:535-586 (no masking), differentiable in
    (vertices, tx, rx).  cand: LongTensor [C, k] of triangle ids (even ids when quads)."""
    tv = vertices[triangles]
    nr = normals(vertices, triangles)
    mv = tv[cand][:, :, 0, :]
    mn = nr[cand]
    a = tx[:, None, None, :]
    b = rx[None, :, None, :]
    Ntx, Nrx, C, k = tx.shape[0], rx.shape[0], cand.shape[0], cand.shape[1]
    a_ = a.expand(Ntx, Nrx, C, 3)
    b_ = b.expand(Ntx, Nrx, C, 3)
    if k:
        inner = image_method(a_, b_, mv.expand(Ntx, Nrx, C, k, 3), mn.expand(Ntx, Nrx, C, k, 3))
        return torch.cat((a_[..., None, :], inner, b_[..., None, :]), dim=-2)
    return torch.cat((a_[..., None, :], b_[..., None, :]), dim=-2)


def differentiable_distance(vertices, triangles, o, d, faces):
    """ME:226-255."""
    tv = vertices[triangles][faces.clamp(min=0)]
    v0, v1, v2 = tv[:, 0], tv[:, 1], tv[:, 2]
    e1, e2 = v1 - v0, v2 - v0
    h = torch.linalg.cross(d, e2)
    a = (h * e1).sum(-1)
    a = torch.where(a == 0.0, torch.full_like(a, float("inf")), a)
    f = 1.0 / a
    s = o - v0
    q = torch.linalg.cross(s, e1)
    t = f * (q * e2).sum(-1)
    return torch.where(faces != -1, t, torch.full_like(t, float("inf")))
