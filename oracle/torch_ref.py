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
    return x - 2.0 * _dot3(inc, n)[..., None] * n  # IM:73-79


def intersection_of_ray_with_plane(o, d, p, n):
    v = p - o
    un = _dot3(d, n)[..., None]
    vn = _dot3(v, n)[..., None]
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


# --------------------------------------------------------------------------------------------
# Smoothed ("soft mask") mode -- SURVEY.md section 8 row f4.  Forward oracle in float32 (explicit
# operation order: 3-term sums associate left to right like the C oracle, so `t` is bit-identical
# to the hard mode) AND gradient oracle through torch.autograd (float64 for the truth).
# Reductions use amin/amax: like jnp.min/jnp.max they propagate NaN and split gradients evenly
# between ties.
# --------------------------------------------------------------------------------------------
def _dot3(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross3(a, b):
    return torch.stack((a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]), dim=-1)


class _Logistic(torch.autograd.Function):
    """`lax.logistic`: value 1/(1+exp(-y)) the way XLA expands it, derivative s(1-s) (JAX's own JVP
    rule, which stays finite where exp(-y) overflows)."""

    @staticmethod
    def forward(ctx, y):
        s = 1.0 / (1.0 + torch.exp(-y))
        ctx.save_for_backward(s)
        return s

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return g * (s * (1.0 - s))


def smoothing_function(x, smoothing_factor=1.0):
    """utils.py:70-89: sigmoid(x * alpha)."""
    return _Logistic.apply(x * smoothing_factor)


def _min_with_one(*terms):
    """`jnp.stack(terms, -1).min(-1, initial=1.0)` (UT:1288-1296)."""
    st = torch.stack(torch.broadcast_tensors(*terms, torch.ones((), dtype=terms[0].dtype)), dim=-1)
    return st.amin(dim=-1)


def ray_intersect_triangle(o, d, tv, *, epsilon, smoothing_factor=None):
    """UT:1262-1322, both modes.  Returns (t, hit); hit is bool (hard) or float (smoothed)."""
    v0, v1, v2 = tv[..., 0, :], tv[..., 1, :], tv[..., 2, :]
    e1, e2 = v1 - v0, v2 - v0
    o, d, e1, e2, v0 = torch.broadcast_tensors(o, d, e1, e2, v0)
    h = _cross3(d, e2)
    a = _dot3(h, e1)
    a = torch.where(a == 0.0, torch.full_like(a, float("inf")), a)
    sf = smoothing_factor
    if sf is not None:
        hit = smoothing_function(a.abs() - epsilon, sf)
    else:
        hit = a.abs() > epsilon
    f = 1.0 / a
    s = o - v0
    u = f * _dot3(s, h)
    if sf is not None:
        hit = _min_with_one(hit, smoothing_function(u - 0.0, sf), smoothing_function(1.0 - u, sf))
    else:
        hit = hit & (u >= 0.0) & (u <= 1.0)
    q = _cross3(s, e1)
    v = f * _dot3(q, d)
    if sf is not None:
        hit = _min_with_one(hit, smoothing_function(v - 0.0, sf), smoothing_function(1.0 - (u + v), sf))
    else:
        hit = hit & (v >= 0.0) & (u + v <= 1.0)
    t = f * _dot3(q, e2)
    if sf is not None:
        hit = torch.minimum(hit, smoothing_function(t - epsilon, sf))
    else:
        hit = hit & (t > epsilon)
    return t, hit


def ray_intersect_any_triangle(o, d, tv, active=None, *, epsilon, hit_tol, smoothing_factor,
                               batch_size=512):
    """UT:1436-1537, smoothed mode: per tile of `batch_size` triangles the sum of
    min(hit, sigmoid((1 - hit_tol - t) alpha)) over the active triangles; tiles are combined with
    `(left + right).clip(max=1)` (UT:1475-1476), remainder tile last (UT:1525-1537)."""
    T = tv.shape[-3]
    batch = torch.broadcast_shapes(o.shape[:-1], d.shape[:-1], tv.shape[:-3],
                                   active.shape[:-1] if active is not None else ())
    acc = torch.zeros(batch, dtype=o.dtype)
    if T == 0:
        return acc
    thr = 1.0 - hit_tol
    bs = T if batch_size is None else max(min(batch_size, T), 1)
    nb, rem = divmod(T, bs)
    tiles = [(i * bs, (i + 1) * bs) for i in range(nb)] + ([(T - rem, T)] if rem else [])
    for lo, hi in tiles:
        t, hit = ray_intersect_triangle(o[..., None, :], d[..., None, :], tv[..., lo:hi, :, :],
                                        epsilon=epsilon, smoothing_factor=smoothing_factor)
        w = torch.minimum(hit, smoothing_function(thr - t, smoothing_factor))
        if active is not None:
            w = torch.where(active[..., lo:hi], w, torch.zeros_like(w))
        acc = (acc + w.sum(dim=-1)).clamp(max=1.0)
    return acc


def consecutive_vertices_are_on_same_side_of_mirror(vertices, mv, mn, *, smoothing_factor=None):
    """IM:440-454."""
    dp = _dot3(vertices[..., :-2, :] - mv, mn)
    dn = _dot3(vertices[..., 2:, :] - mv, mn)
    if smoothing_factor is not None:
        return smoothing_function(torch.sign(dp) * torch.sign(dn), smoothing_factor)
    return torch.sign(dp) == torch.sign(dn)


def trace_smooth(vertices, triangles, tx, rx, cand, *, mask=None, assume_quads=False, epsilon, hit_tol,
                 min_len, smoothing_factor, batch_size=512):
    """SV:499-770 with `smoothing_factor` set: returns (full path vertices [Ntx,Nrx,C,k+2,3],
    soft mask [Ntx,Nrx,C]), differentiable in (vertices, tx, rx).  Candidate rows holding ids outside
    [0, T) (the -1 padding of SV:912-918) get mask 0 and zero vertices, like the hard oracle."""
    sf = smoothing_factor
    T = triangles.shape[0]
    C, k = cand.shape
    Ntx, Nrx = tx.shape[0], rx.shape[0]
    ok_rows = ((cand >= 0) & (cand + (1 if assume_quads else 0) < T)).all(dim=-1) if k else \
        torch.ones(C, dtype=torch.bool)
    candc = torch.where(ok_rows[:, None], cand, torch.zeros_like(cand))
    full = trace_vertices(vertices, triangles, tx, rx, candc)
    tv = vertices[triangles]
    nr = normals(vertices, triangles)
    ro = full[..., :-1, :]
    rd = full[..., 1:, :] - full[..., :-1, :]
    one = torch.ones((), dtype=full.dtype)
    if k:
        if assume_quads:
            pair = torch.stack((candc, candc + 1), dim=-1)                 # [C,k,2]
            _, h = ray_intersect_triangle(ro[..., :-1, None, :], rd[..., :-1, None, :], tv[pair],
                                          epsilon=epsilon, smoothing_factor=sf)     # [..,C,k,2]
            h = torch.cat((h, torch.zeros_like(h[..., :1])), dim=-1).amax(dim=-1)  # max, initial 0
        else:
            _, h = ray_intersect_triangle(ro[..., :-1, :], rd[..., :-1, :], tv[candc],
                                          epsilon=epsilon, smoothing_factor=sf)
        inside = torch.cat((h, one.expand(*h.shape[:-1], 1)), dim=-1).amin(dim=-1)
        ss = consecutive_vertices_are_on_same_side_of_mirror(full, tv[candc][:, :, 0, :], nr[candc],
                                                             smoothing_factor=sf)
        valid = torch.cat((ss, one.expand(*ss.shape[:-1], 1)), dim=-1).amin(dim=-1)
    else:
        inside = one.expand(Ntx, Nrx, C)
        valid = one.expand(Ntx, Nrx, C)
    b = ray_intersect_any_triangle(ro, rd, tv, mask, epsilon=epsilon, hit_tol=hit_tol,
                                   smoothing_factor=sf, batch_size=batch_size)      # [..,C,k+1]
    blocked = torch.cat((b, torch.zeros_like(b[..., :1])), dim=-1).amax(dim=-1)
    len2 = _dot3(rd, rd)
    ts = smoothing_function(min_len - len2, sf)
    too_small = torch.cat((ts, torch.zeros_like(ts[..., :1])), dim=-1).amax(dim=-1)
    finite = torch.isfinite(full).all(dim=-1).all(dim=-1)
    full = torch.where(finite[..., None, None], full, torch.zeros_like(full))
    m = torch.stack((inside, valid, 1.0 - blocked, 1.0 - too_small, finite.to(full.dtype),
                     one.expand(Ntx, Nrx, C)), dim=-1).amin(dim=-1)
    if mask is not None and k:
        act = mask[candc].all(dim=-1)
        if assume_quads:
            act = act & mask[candc + 1].all(dim=-1)
        m = m * act.to(full.dtype)
    m = torch.where(ok_rows, m, torch.zeros_like(m))
    full = torch.where(ok_rows[:, None, None], full, torch.zeros_like(full))
    return full, m
