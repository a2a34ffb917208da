"""GPU parity: HIP ray/triangle operators (through the C ABI) vs the CPU oracle.

Bar: bit-exact hit masks and hit indices; `t` bit-exact too (same operation order, no FMA).
Mirrors differt/tests/geometry/test_utils.py:555-714, 910-962 of the reference.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


def test_known_answers(G, goldens):
    """test_utils.py:555-606."""
    g = goldens["ray_intersect_triangle_t_and_hit"]
    o = np.asarray(g["ray_origin"], np.float32)
    d = np.asarray(g["ray_directions"], np.float32)
    tv = np.asarray(g["triangle_vertices"], np.float32)
    t, hit = G.ray_intersect_triangle(o[None, None, :], d[:, None, :], tv)
    np.testing.assert_array_equal(_np(t), np.asarray(g["expected_t"], np.float32))
    np.testing.assert_array_equal(_np(hit), np.asarray(g["expected_hit"]))
    g = goldens["ray_intersect_triangle_hits"]
    tri = np.asarray([g["triangle"]], np.float32)
    for case in g["cases"]:
        o = np.asarray(case["orig"], np.float32)
        dd = np.asarray(case["dest"], np.float32) - o
        t, hit = G.ray_intersect_triangle(o, dd, tri)
        assert bool(_np((t < 1.0) & hit)[0]) == case["expected"]


@pytest.mark.parametrize("B,R,T,shared", [(64, 256, 10000, True), (64, 256, 10000, False), (5, 33, 1023, False),
                                          (3, 40, 2064, False), (7, 9, 16, True), (2, 300, 1028, False)])
def test_dense_batched_equals_single_launches(G, rng, B, R, T, shared):
    """drt_ray_intersect_triangle_dense_batched (configs[1] under leading batch axes,
    docs/source/batch_axes.md:67-80): B problems in one launch, bit-identical to B single launches and
    to the oracle, with shared and with per-problem triangle sets, on the aligned and the fallback kernels."""
    o = (rng.uniform(-1, 1, (B, R, 3)) * 50).astype(np.float32)
    d = (rng.uniform(-1, 1, (B, R, 3)) * 50).astype(np.float32) - o
    nb = 1 if shared else B
    c = (rng.uniform(-1, 1, (nb, T, 1, 3)) * 50).astype(np.float32)
    tv = (c + np.concatenate([np.zeros((nb, T, 1, 3)), rng.normal(size=(nb, T, 2, 3)) * 2], axis=2)).astype(np.float32)
    t, hit = G.ray_intersect_triangle(o[:, :, None, :], d[:, :, None, :], tv[:, None])
    assert t.shape == (B, R, T) and hit.shape == (B, R, T)
    for b in sorted({0, B // 2, B - 1}):
        tb = tv[0 if shared else b]
        t1, h1 = G.ray_intersect_triangle(o[b][:, None, :], d[b][:, None, :], tb)
        assert torch.equal(t[b].view(torch.int32), t1.view(torch.int32)) and torch.equal(hit[b], h1)
        et, eh = orc.ray_intersect_triangle_dense(o[b], d[b], tb)
        np.testing.assert_array_equal(_np(hit[b]), eh)
        np.testing.assert_array_equal(_np(t[b]).view(np.uint32), et.view(np.uint32))


@pytest.mark.parametrize("R,T", [(1, 1), (7, 5), (256, 10000), (33, 1023), (64, 4099), (300, 1028), (3, 16), (5, 1040),
                                 (40, 2064), (2000, 1024)])
def test_dense_bit_exact(G, rng, R, T):
    """cfg2 shape (256 x 10000) and ragged sizes (T % 4 != 0 -> scalar-store path)."""
    o = (rng.uniform(-1, 1, (R, 3)) * 50).astype(np.float32)
    d = (rng.uniform(-1, 1, (R, 3)) * 50).astype(np.float32) - o
    c = (rng.uniform(-1, 1, (T, 1, 3)) * 50).astype(np.float32)
    tv = (c + np.concatenate([np.zeros((T, 1, 3)), rng.normal(size=(T, 2, 3)) * 2], axis=1)).astype(np.float32)
    et, eh = orc.ray_intersect_triangle_dense(o, d, tv)
    t, hit = G.ray_intersect_triangle(o[:, None, :], d[:, None, :], tv)
    assert t.shape == (R, T)
    np.testing.assert_array_equal(_np(hit), eh)
    np.testing.assert_array_equal(_np(t).view(np.uint32), et.view(np.uint32))


def test_dense_unaligned_triangle_pointer(G, rng):
    """A triangle array that starts 4 bytes off a 16-B boundary takes the direct-load path of the aligned
    kernel (no LDS staging of the triangles): same bits."""
    R, T = 16, 2048
    flat = torch.as_tensor(rng.normal(size=9 * T + 1).astype(np.float32) * 3, device="cuda")
    tv = flat[1:].view(T, 3, 3)
    assert tv.data_ptr() % 16 == 4 and tv.is_contiguous()
    o = rng.normal(size=(R, 3)).astype(np.float32) * 5
    d = rng.normal(size=(R, 3)).astype(np.float32)
    et, eh = orc.ray_intersect_triangle_dense(o, d, _np(tv))
    t, hit = G.ray_intersect_triangle(torch.as_tensor(o, device="cuda")[:, None, :],
                                      torch.as_tensor(d, device="cuda")[:, None, :], tv)
    np.testing.assert_array_equal(_np(hit), eh)
    np.testing.assert_array_equal(_np(t).view(np.uint32), et.view(np.uint32))


@pytest.mark.parametrize("epsilon", [0.0, 1e-40, 1.1754944e-38, 1e-3, 0.5])
def test_dense_epsilon_range(G, rng, epsilon):
    """The fast path folds the reference's `|a| > eps` into its range check (every |a| of the wave must
    exceed max(eps, largest denormal), geom.hpp `mt_fast_threshold`): both sides of the 2^-126 threshold, and
    epsilons large enough that waves mix determinants above and below them, stay bit-exact."""
    R, T = 48, 2048
    tv = rng.normal(size=(T, 3, 3)).astype(np.float32) * 3
    cen = tv.mean(axis=1)
    o = (cen[:R] + rng.normal(size=(R, 3)) * 5).astype(np.float32)
    d = ((cen[:R] - o) * 2).astype(np.float32)
    et, eh = orc.ray_intersect_triangle_dense(o, d, tv, epsilon=epsilon)
    t, hit = G.ray_intersect_triangle(o[:, None, :], d[:, None, :], tv, epsilon=epsilon)
    assert eh.sum() > 0 and not eh.all()
    np.testing.assert_array_equal(_np(hit), eh)
    np.testing.assert_array_equal(_np(t).view(np.uint32), et.view(np.uint32))


def test_dense_hits_present(G, rng):
    """Make sure the parity above is not vacuous: rays aimed at triangle centroids do hit."""
    T = 2048
    tv = rng.normal(size=(T, 3, 3)).astype(np.float32) * 3
    cen = tv.mean(axis=1)
    o = cen + rng.normal(size=(T, 3)).astype(np.float32) * 5
    d = (cen - o) * np.float32(2.0)
    et, eh = orc.ray_intersect_triangle_dense(o[:64], d[:64], tv)
    t, hit = G.ray_intersect_triangle(o[:64, None, :], d[:64, None, :], tv)
    assert eh.sum() >= 32
    np.testing.assert_array_equal(_np(hit), eh)
    np.testing.assert_array_equal(_np(t).view(np.uint32), et.view(np.uint32))


@pytest.mark.parametrize("shapes", [((3,), (3,), (3, 3)), ((15, 5, 3), (15, 5, 3), (5, 3, 3)), ((4, 1, 3), (1, 6, 3), (4, 6, 3, 3))])
def test_paired_broadcast(G, rng, shapes):
    """test_utils.py:609-646."""
    so, sd, st = shapes
    o = rng.normal(size=so).astype(np.float32)
    d = rng.normal(size=sd).astype(np.float32)
    tv = rng.normal(size=st).astype(np.float32)
    et, eh = orc.ray_intersect_triangle(o, d, tv)
    t, hit = G.ray_intersect_triangle(o, d, tv)
    assert tuple(t.shape) == et.shape
    np.testing.assert_array_equal(_np(hit), eh)
    np.testing.assert_array_equal(_np(t).view(np.uint32), et.view(np.uint32))
    assert (_np(t)[_np(hit)] > 0).all()


@pytest.mark.parametrize("epsilon", [None, 1e-6, 1e-2])
@pytest.mark.parametrize("hit_tol", [None, 0.0, 0.001, -0.5, 0.5])
@pytest.mark.parametrize("with_active", [True, False])
@pytest.mark.parametrize(
    "shapes",
    [((20, 10, 3), (20, 10, 3), (20, 10, 5, 3, 3)), ((10, 3), (10, 3), (1, 3, 3)), ((3,), (3,), (1, 3, 3)),
     ((700, 3), (700, 3), (777, 3, 3))],
)
def test_any_triangle(G, rng, shapes, epsilon, hit_tol, with_active):
    """test_utils.py:649-714 + a shared-triangle case with random active mask and real hits."""
    so, sd, st = shapes
    o = rng.normal(size=so).astype(np.float32)
    d = rng.normal(size=sd).astype(np.float32) * 3
    tv = rng.normal(size=st).astype(np.float32)
    act = (rng.random(st[:-2]) > 0.3) if with_active else None
    exp = orc.ray_intersect_any_triangle(o, d, tv, act, epsilon=epsilon, hit_tol=hit_tol)
    got = G.ray_intersect_any_triangle(o, d, tv, act, epsilon=epsilon, hit_tol=hit_tol, batch_size=11)
    assert tuple(got.shape) == exp.shape
    np.testing.assert_array_equal(_np(got), exp)


@pytest.mark.parametrize("epsilon", [None, 1e-2])
@pytest.mark.parametrize("with_active", [True, False])
@pytest.mark.parametrize("batch_size", [11, 512, None])
@pytest.mark.parametrize(
    "shapes",
    [((10, 3), (1, 3), (30, 3, 3)), ((100, 3), (100, 3), (1, 300, 3, 3)), ((4, 3), (4, 3), (0, 3, 3)),
     ((5, 3), (5, 3), (5, 40, 3, 3)), ((1000, 3), (1000, 3), (1500, 3, 3))],
)
def test_first_triangle_hit(G, rng, shapes, epsilon, with_active, batch_size):
    """test_utils.py:910-962, with index equality (the reference leaves it commented out)."""
    so, sd, st = shapes
    o = rng.normal(size=so).astype(np.float32)
    d = rng.normal(size=sd).astype(np.float32) * 3
    tv = rng.normal(size=st).astype(np.float32)
    act = (rng.random(st[:-2]) > 0.3) if with_active else None
    ei, et = orc.first_triangle_hit_by_ray(o, d, tv, act, batch_size=batch_size, epsilon=epsilon)
    gi, gt = G.first_triangle_hit_by_ray(o, d, tv, act, batch_size=batch_size, epsilon=epsilon)
    np.testing.assert_array_equal(_np(gi), ei)
    np.testing.assert_array_equal(_np(gt).view(np.uint32), et.view(np.uint32))
    if st[-3] > 0:
        assert (ei >= 0).any()


@pytest.mark.parametrize("batch_size,expected", [(4, 4), (6, 0), (2, 4), (512, 0), (1, 5)])
def test_first_hit_tie_break(G, batch_size, expected):
    """_utils.py:1865-1867, 1886: duplicate triangles."""
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tv = np.stack([tri] * 6)
    o = np.array([[0.25, 0.25, 1.0]], np.float32)
    d = np.array([[0.0, 0.0, -1.0]], np.float32)
    ei, _ = orc.first_triangle_hit_by_ray(o, d, tv, batch_size=batch_size)
    gi, gt = G.first_triangle_hit_by_ray(o, d, tv, batch_size=batch_size)
    assert int(ei[0]) == expected and int(_np(gi)[0]) == expected and float(_np(gt)[0]) == 1.0


def test_tie_break_across_splits(G, rng):
    """Many duplicated triangles spread over several LDS tiles / blockIdx.y splits."""
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    T = 5000
    tv = rng.normal(size=(T, 3, 3)).astype(np.float32) + 10
    dup = rng.choice(T, size=40, replace=False)
    tv[dup] = tri
    o = np.tile(np.array([[0.25, 0.25, 1.0]], np.float32), (3, 1))
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (3, 1))
    for bs in (512, 100, 7):
        ei, et = orc.first_triangle_hit_by_ray(o, d, tv, batch_size=bs)
        gi, gt = G.first_triangle_hit_by_ray(o, d, tv, batch_size=bs)
        np.testing.assert_array_equal(_np(gi), ei)
        np.testing.assert_array_equal(_np(gt), et)


def test_empty(G):
    """_utils.py:1441-1450, 1848-1857."""
    o = np.zeros((4, 3), np.float32)
    d = np.ones((4, 3), np.float32)
    tv = np.zeros((0, 3, 3), np.float32)
    assert not _np(G.ray_intersect_any_triangle(o, d, tv)).any()
    i, t = G.first_triangle_hit_by_ray(o, d, tv)
    assert (_np(i) == -1).all() and np.isinf(_np(t)).all()
    t, h = G.ray_intersect_triangle(np.zeros((0, 1, 3), np.float32), np.zeros((0, 1, 3), np.float32), np.zeros((5, 3, 3), np.float32))
    assert t.shape == (0, 5)


# ------------------------------------------------------------------ hard-mode t is differentiable ----
def _f64_grads(fn, *xs):
    ys = [torch.tensor(np.asarray(x, np.float64), requires_grad=True) for x in xs]
    fn(*ys).backward()
    return [y.grad.numpy() for y in ys]


def _close(got, exp, rtol=1e-5):
    scale = float(np.abs(exp).max()) + 1e-30
    assert float(np.abs(got - exp).max()) <= rtol * scale, (float(np.abs(got - exp).max()), scale)


@pytest.mark.parametrize("shape", ["dense", "paired", "broadcast"])
def test_ray_intersect_triangle_t_gradients(shape):
    """Reference: t of ray_intersect_triangle is plain JAX arithmetic (_utils.py:1316), so jax.grad
    flows through it; here drt_ray_intersect_triangle_vjp, checked against float64 autograd of the
    torch restatement (<= 1e-5 rel of the largest entry).  Forward values stay bit-identical."""
    import differt_amd.geometry as G
    from oracle import torch_ref

    rng = np.random.default_rng(17)
    R, T = 37, 53
    eps = 10 * 1.1920929e-7
    # well-conditioned pairs (roughly horizontal triangles, rays pointing down within ~35 degrees): with
    # near-parallel ray / triangle pairs 1/a is huge and float32 itself is 1e-4 off the float64 truth
    def flat_triangles(num, size, span):
        # no slivers: e1 ~ (+x), e2 ~ (+y), small tilt -> |a| = |d . (e1 x e2)| stays O(|d| size^2)
        v0 = rng.uniform(-span, span, (num, 3)) * np.array([1.0, 1.0, 0.3])
        e1 = np.column_stack((rng.uniform(1, 2, num), rng.uniform(-0.3, 0.3, num), rng.uniform(-0.25, 0.25, num))) * size
        e2 = np.column_stack((rng.uniform(-0.3, 0.3, num), rng.uniform(1, 2, num), rng.uniform(-0.25, 0.25, num))) * size
        return np.stack((v0, v0 + e1, v0 + e2), axis=1).astype(np.float32)

    tvn = flat_triangles(T, 1.0, 5.0)
    on = np.column_stack((rng.uniform(-4, 4, R), rng.uniform(-4, 4, R), rng.uniform(8, 12, R))).astype(np.float32)
    dn = np.column_stack((rng.uniform(-4, 4, (R, 2)), -rng.uniform(8, 12, R))).astype(np.float32)
    if shape == "dense":
        args = (on[:, None, :], dn[:, None, :], tvn)
    elif shape == "paired":
        args = (on, dn, tvn[rng.integers(0, T, R)])
    else:  # one origin broadcast against [R] directions and [R] triangles
        args = (on[:1], dn, tvn[rng.integers(0, T, R)])
    w = rng.normal(size=np.broadcast_shapes(args[0].shape[:-1], args[1].shape[:-1], args[2].shape[:-2])).astype(np.float32)

    o, d, tv = (torch.tensor(a, device="cuda", requires_grad=True) for a in args)
    t, hit = G.ray_intersect_triangle(o, d, tv)
    t0, hit0 = G.ray_intersect_triangle(o.detach(), d.detach(), tv.detach())
    assert torch.equal(t.detach().view(torch.int32), t0.view(torch.int32)) and torch.equal(hit, hit0)
    assert hit.dtype == torch.bool and not hit.requires_grad
    (t * torch.tensor(w, device="cuda")).sum().backward()

    def ref(o64, d64, tv64):
        tt, _ = torch_ref.ray_intersect_triangle(o64, d64, tv64, epsilon=eps)
        return (tt * torch.tensor(w, dtype=torch.float64)).sum()

    for got, exp in zip((o.grad, d.grad, tv.grad), _f64_grads(ref, *args)):
        _close(got.cpu().numpy(), exp)


def test_ray_intersect_triangle_t_gradient_degenerate_and_zero_cotangent():
    """a == 0 (ray parallel to the plane): the reference turns a into a constant inf, f = 0 -> zero
    gradient, no NaN; a zero cotangent never produces 0 * inf."""
    import differt_amd.geometry as G

    tv = torch.tensor([[[0.0, 0, 0], [1, 0, 0], [0, 1, 0]]], device="cuda", requires_grad=True)
    o = torch.tensor([[0.2, 0.2, 1.0], [0.2, 0.2, 1.0]], device="cuda", requires_grad=True)
    d = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0]], device="cuda", requires_grad=True)  # first one is parallel
    t, hit = G.ray_intersect_triangle(o, d, tv.expand(2, 3, 3))
    assert hit.tolist() == [False, True]
    (t * torch.tensor([1.0, 0.0], device="cuda")).sum().backward()
    for g in (o.grad, d.grad, tv.grad):
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) == 0.0


def test_first_triangle_hit_by_ray_free_function_t_gradient():
    """_utils.py:1775-1960: the free operator's t is differentiable too (min over the hit distances)."""
    import differt_amd.geometry as G
    from oracle import torch_ref

    rng = np.random.default_rng(23)
    R, T = 64, 200
    eps = 10 * 1.1920929e-7
    v0 = rng.uniform(-20, 20, (T, 3)) * np.array([1.0, 1.0, 0.2])
    e1 = np.column_stack((rng.uniform(3, 6, T), rng.uniform(-1, 1, T), rng.uniform(-0.5, 0.5, T)))
    e2 = np.column_stack((rng.uniform(-1, 1, T), rng.uniform(3, 6, T), rng.uniform(-0.5, 0.5, T)))
    tvn = np.stack((v0, v0 + e1, v0 + e2), axis=1).astype(np.float32)
    on = np.column_stack((rng.uniform(-15, 15, R), rng.uniform(-15, 15, R), rng.uniform(20, 30, R))).astype(np.float32)
    tgt = tvn.mean(axis=1)[rng.integers(0, T, R)]
    dn = (tgt - on).astype(np.float32)
    o, d, tv = (torch.tensor(a, device="cuda", requires_grad=True) for a in (on, dn, tvn))
    idx, t = G.first_triangle_hit_by_ray(o, d, tv)
    idx0, t0 = G.first_triangle_hit_by_ray(o.detach(), d.detach(), tv.detach())
    assert torch.equal(idx, idx0) and torch.equal(t.detach().view(torch.int32), t0.view(torch.int32))
    hitm = (idx >= 0).cpu().numpy()
    assert hitm.sum() > R // 2
    w = rng.normal(size=R).astype(np.float32)
    torch.where(idx >= 0, t * torch.tensor(w, device="cuda"), torch.zeros_like(t)).sum().backward()
    sel = idx.clamp(min=0).long().cpu().numpy()

    def ref(o64, d64, tv64):
        tt, _ = torch_ref.ray_intersect_triangle(o64, d64, tv64[torch.as_tensor(sel)], epsilon=eps)
        return (tt * torch.tensor(w * hitm, dtype=torch.float64)).sum()

    for got, exp in zip((o.grad, d.grad, tv.grad), _f64_grads(ref, on, dn, tvn)):
        _close(got.cpu().numpy(), exp)
