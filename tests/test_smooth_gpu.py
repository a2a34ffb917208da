"""GPU parity of the smoothed ("soft mask") mode through the C ABI vs the torch restatement.

Bar: `t` and path vertices bit-identical to the hard mode; confidences within 1e-5 relative
(+1e-6 absolute: they live in [0, 1] and come out of expf) of the float32 restatement
(oracle/torch_ref.py, pinned by tests/test_smooth_oracle.py); gradients within 1e-5 of float64
torch.autograd, or no worse than 4x plain float32 autograd where the sample is ill-conditioned.
Mirrors the reference's "large smoothing factor matches no smoothing" tests
(differt/tests/geometry/test_utils.py:636-646, 701-715; test_image_method.py:246-255;
test_scene.py:366-442).
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc
from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu

EPS, TOL, MINLEN = orc.DEFAULT_EPSILON, orc.DEFAULT_HIT_TOL, orc.DEFAULT_MIN_LEN
RTOL, ATOL = 1e-5, 1e-6


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _t(x, dtype=torch.float32, **kw):
    return torch.tensor(np.asarray(x), dtype=dtype, **kw)


def _check_grads(got, ref32, truth, names):
    for g, r, t, name in zip(got, ref32, truth, names):
        g, r, t = (np.asarray(x, np.float64).reshape(-1) for x in (g, r, t))
        scale = np.abs(t).max() + 1e-30
        err_gpu = np.abs(g - t).max() / scale
        err_ref = np.abs(r - t).max() / scale
        assert np.isfinite(g).all(), name
        assert err_gpu <= max(RTOL, 4 * err_ref), (name, err_gpu, err_ref)


# ------------------------------------------------------------------ Moller-Trumbore ----
SHAPES = [((50, 1, 3), (1, 40, 3, 3)), ((33, 3), (33, 3, 3)), ((4, 1, 3), (4, 9, 3, 3)), ((3,), (3, 3))]


@pytest.mark.parametrize("shape", SHAPES)
def test_mt_large_smoothing_matches_hard(G, rng, shape):
    """tests/geometry/test_utils.py:609-646."""
    o = rng.uniform(-1, 1, shape[0]).astype(np.float32)
    d = rng.uniform(-1, 1, shape[0]).astype(np.float32)
    tv = rng.uniform(-1, 1, shape[1]).astype(np.float32)
    t_hard, hit_hard = G.ray_intersect_triangle(o, d, tv)
    t, hit = G.ray_intersect_triangle(o, d, tv, smoothing_factor=1e8)
    assert hit.dtype == torch.float32
    np.testing.assert_array_equal(_bits(_np(t)), _bits(_np(t_hard)))
    np.testing.assert_array_equal(_np(hit) > 0.5, _np(hit_hard))
    to, ho = orc.ray_intersect_triangle(o, d, tv)
    np.testing.assert_array_equal(_bits(_np(t)), _bits(to))
    np.testing.assert_array_equal(_np(hit) > 0.5, ho)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("sf", [0.5, 10.0, 1000.0])
def test_mt_smooth_vs_restatement(G, rng, shape, sf):
    o = rng.uniform(-1, 1, shape[0]).astype(np.float32)
    d = rng.uniform(-1, 1, shape[0]).astype(np.float32)
    tv = rng.uniform(-1, 1, shape[1]).astype(np.float32)
    t, hit = G.ray_intersect_triangle(o, d, tv, smoothing_factor=sf)
    te, he = tr.ray_intersect_triangle(_t(o), _t(d), _t(tv), epsilon=EPS, smoothing_factor=sf)
    np.testing.assert_array_equal(_bits(_np(t)), _bits(te.numpy()))
    np.testing.assert_allclose(_np(hit), he.numpy(), rtol=RTOL, atol=ATOL)
    assert 0.0 <= float(hit.min()) and float(hit.max()) <= 1.0


def test_mt_smooth_degenerate(G):
    """a == 0 (ray in the triangle plane) and infinities: no NaN unless the inputs carry one."""
    tv = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    o = np.array([[0.2, 0.2, 0.0], [0.2, 0.2, 1.0], [np.inf, 0, 0]], np.float32)
    d = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0, 0, 1]], np.float32)
    t, hit = G.ray_intersect_triangle(o[:, None], d[:, None], tv[None], smoothing_factor=7.0)
    te, he = tr.ray_intersect_triangle(_t(o)[:, None], _t(d)[:, None], _t(tv)[None], epsilon=EPS, smoothing_factor=7.0)
    np.testing.assert_array_equal(np.isnan(_np(hit)), np.isnan(he.numpy()))
    np.testing.assert_allclose(_np(hit), he.numpy(), rtol=RTOL, atol=ATOL)
    assert float(hit[0, 0]) <= 0.5 and float(hit[1, 0]) > 0.75


@pytest.mark.parametrize("shape", SHAPES[:3])
@pytest.mark.parametrize("sf", [2.0, 30.0])
def test_mt_smooth_vjp(G, rng, shape, sf):
    o = rng.uniform(-1, 1, shape[0])
    d = rng.uniform(-1, 1, shape[0])
    tv = rng.uniform(-1, 1, shape[1])
    batch = np.broadcast_shapes(shape[0][:-1], shape[1][:-2])
    wt, wh = rng.normal(size=batch) * 1e-2, rng.normal(size=batch)

    def grads(dtype, device, fn):
        ins = [torch.tensor(x, dtype=dtype, device=device, requires_grad=True) for x in (o, d, tv)]
        t, hit = fn(*ins)
        t = torch.where(torch.isfinite(t) & (t.abs() < 1e3), t, torch.zeros_like(t))
        (t * torch.tensor(wt, dtype=dtype, device=device) + hit * torch.tensor(wh, dtype=dtype, device=device)).sum().backward()
        return [x.grad.detach().cpu().numpy() for x in ins]

    ref = lambda a, b, c: tr.ray_intersect_triangle(a, b, c, epsilon=EPS, smoothing_factor=sf)  # noqa: E731
    g64 = grads(torch.float64, "cpu", ref)
    g32 = grads(torch.float32, "cpu", ref)
    gg = grads(torch.float32, "cuda", lambda a, b, c: G.ray_intersect_triangle(a, b, c, smoothing_factor=sf))
    assert all(np.abs(x).max() > 0 for x in g64)
    _check_grads(gg, g32, g64, ("origins", "directions", "triangle_vertices"))


# ------------------------------------------------------------------ any triangle ----
def _any_case(rng, per_ray_tv, with_active, R=40, T=37):
    o = rng.uniform(-1, 1, (R, 3)).astype(np.float32)
    d = (rng.uniform(-1, 1, (R, 3)) * 3).astype(np.float32)
    tv = rng.uniform(-1, 1, ((R, T, 3, 3) if per_ray_tv else (T, 3, 3))).astype(np.float32)
    act = None
    if with_active:
        act = rng.random((R, T) if per_ray_tv else (T,)) > 0.3
    return o, d, tv, act


@pytest.mark.parametrize("per_ray_tv", [False, True])
@pytest.mark.parametrize("with_active", [False, True])
def test_any_large_smoothing_matches_hard(G, rng, per_ray_tv, with_active):
    """tests/geometry/test_utils.py:649-715, batch_size=11 leaves a remainder tile."""
    o, d, tv, act = _any_case(rng, per_ray_tv, with_active)
    hard = G.ray_intersect_any_triangle(o, d, tv, act)
    got = G.ray_intersect_any_triangle(o, d, tv, act, smoothing_factor=1e8, batch_size=11)
    assert got.dtype == torch.float32
    np.testing.assert_array_equal(_np(got) > 0.5, _np(hard))
    np.testing.assert_array_equal(_np(hard), orc.ray_intersect_any_triangle(o, d, tv, act))
    assert _np(hard).any() and not _np(hard).all()
    empty = G.ray_intersect_any_triangle(o, d, tv[..., :0, :, :], smoothing_factor=3.0)
    assert empty.dtype == torch.float32 and tuple(empty.shape) == (40,) and not bool(empty.any())


@pytest.mark.parametrize("per_ray_tv", [False, True])
@pytest.mark.parametrize("with_active", [False, True])
@pytest.mark.parametrize(("sf", "batch_size"), [(1.0, 512), (8.0, 11), (40.0, None), (300.0, 1)])
def test_any_smooth_vs_restatement(G, rng, per_ray_tv, with_active, sf, batch_size):
    o, d, tv, act = _any_case(rng, per_ray_tv, with_active)
    got = G.ray_intersect_any_triangle(o, d, tv, act, smoothing_factor=sf, batch_size=batch_size)
    exp = tr.ray_intersect_any_triangle(_t(o), _t(d), _t(tv), None if act is None else torch.tensor(act),
                                        epsilon=EPS, hit_tol=TOL, smoothing_factor=sf, batch_size=batch_size)
    np.testing.assert_allclose(_np(got), exp.numpy(), rtol=RTOL, atol=ATOL)
    assert float(got.max()) <= 1.0


@pytest.mark.parametrize("per_ray_tv", [False, True])
@pytest.mark.parametrize("with_active", [False, True])
def test_any_smooth_vjp(G, rng, per_ray_tv, with_active):
    o, d, tv, act = _any_case(rng, per_ray_tv, with_active, R=24, T=13)
    tv = tv * 0.5  # small triangles: most sums stay below the clip
    w = rng.normal(size=(24,))
    sf = 6.0

    def grads(dtype, device, fn):
        ins = [torch.tensor(x, dtype=dtype, device=device, requires_grad=True) for x in (o, d, tv)]
        out = fn(*ins)
        (out * torch.tensor(w, dtype=dtype, device=device)).sum().backward()
        return out.detach().cpu().numpy(), [x.grad.detach().cpu().numpy() for x in ins]

    actt = None if act is None else torch.tensor(act)
    ref = lambda a, b, c: tr.ray_intersect_any_triangle(a, b, c, actt, epsilon=EPS, hit_tol=TOL,  # noqa: E731
                                                        smoothing_factor=sf, batch_size=5)
    v64, g64 = grads(torch.float64, "cpu", ref)
    _, g32 = grads(torch.float32, "cpu", ref)
    _, gg = grads(torch.float32, "cuda",
                  lambda a, b, c: G.ray_intersect_any_triangle(a, b, c, act, smoothing_factor=sf, batch_size=5))
    assert (v64 < 1.0).any() and (v64 >= 1.0).any()  # both clipped (constant) and unclipped rays
    _check_grads(gg, g32, g64, ("origins", "directions", "triangle_vertices"))


# ------------------------------------------------------------------ same side ----
def test_same_side_smooth(G, rng):
    """tests/geometry/test_image_method.py:222-255."""
    v = rng.normal(size=(10, 6, 3)).astype(np.float32)
    mv = rng.normal(size=(4, 3)).astype(np.float32)
    mn = rng.normal(size=(4, 3)).astype(np.float32)
    hard = G.consecutive_vertices_are_on_same_side_of_mirror(v, mv, mn)
    got = G.consecutive_vertices_are_on_same_side_of_mirror(v, mv, mn, smoothing_factor=1e8)
    assert got.dtype == torch.float32 and tuple(got.shape) == (10, 4)
    np.testing.assert_array_equal(_np(got) > 0.5, _np(hard))
    for sf in (0.3, 5.0):
        got = G.consecutive_vertices_are_on_same_side_of_mirror(v, mv, mn, smoothing_factor=sf)
        exp = tr.consecutive_vertices_are_on_same_side_of_mirror(_t(v), _t(mv), _t(mn), smoothing_factor=sf)
        np.testing.assert_allclose(_np(got), exp.numpy(), rtol=RTOL, atol=ATOL)
    with pytest.raises(TypeError):
        G.consecutive_vertices_are_on_same_side_of_mirror(v[:, :5], mv, mn, smoothing_factor=1.0)


# ------------------------------------------------------------------ tracer ----
def _ref_trace(V, Tr, tx, rx, cand, mask, assume_quads, sf, batch_size=512, dtype=torch.float32):
    return tr.trace_smooth(_t(V, dtype), torch.tensor(Tr, dtype=torch.long), _t(tx, dtype).reshape(-1, 3),
                           _t(rx, dtype).reshape(-1, 3), torch.tensor(np.asarray(cand, np.int64)),
                           mask=None if mask is None else torch.tensor(mask), assume_quads=assume_quads,
                           epsilon=EPS, hit_tol=TOL, min_len=MINLEN, smoothing_factor=sf, batch_size=batch_size)


def _assert_masks_close(got, exp):
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    np.testing.assert_allclose(got[ok], exp[ok], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("mesh_mask", [False, True])
@pytest.mark.parametrize("sf", [1.0, 20.0, 1000.0, 1e8])
def test_trace_smooth_two_buildings(G, goldens, two_buildings, order, assume_quads, mesh_mask, sf):
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    rng = np.random.default_rng(5)
    mask = (rng.random(Tr.shape[0]) > 0.2) if mesh_mask else None
    if mask is not None and assume_quads:
        mask[1::2] = mask[0::2]
    scene = G.Scene(np.asarray(g["tx"], np.float32), np.asarray(g["rx"], np.float32),
                    G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads))
    got = scene.trace_paths(order, solver=G.ExhaustivePathTracer(smoothing_factor=sf))
    hard = scene.trace_paths(order)
    assert got.mask.dtype == torch.float32 and got.mask.shape == hard.mask.shape
    np.testing.assert_array_equal(_np(got.objects), _np(hard.objects))
    np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(_np(hard.vertices)))
    cand = _np(got.objects)[:, 1:-1]
    full, m = _ref_trace(V, Tr, g["tx"], g["rx"], cand, mask, assume_quads, sf)
    _assert_masks_close(_np(got.mask), m.numpy().reshape(-1))
    np.testing.assert_allclose(_np(got.vertices), full.numpy().reshape(-1, order + 2, 3), rtol=1e-5, atol=1e-5)
    if sf == 1e8:  # test_scene.py:366-442: thresholded confidences == hard masks.  The slope must
        # dominate 1/hit_tol: at alpha = 1000 the mirror itself, met at t ~ 1, still "blocks" by ~0.5
        np.testing.assert_array_equal(_np(got.mask) >= 0.5, _np(hard.mask))
        assert torch.equal(got.masked_objects, hard.masked_objects)
        assert int(got.num_valid_paths) == int(hard.num_valid_paths) and (mesh_mask or int(got.num_valid_paths) == 1)


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("batch_size", [512, 7, None])
def test_trace_smooth_canyon(G, rng, order, assume_quads, batch_size):
    """3 TX x 5 RX, masked canyon, padded candidate rows, tile sizes with a remainder."""
    from conftest import canyon_case

    V, Tr, mask, tx, rx, cand = canyon_case(rng, order, assume_quads)
    sel = rng.choice(cand.shape[0], min(150, cand.shape[0]), replace=False)
    cand = np.concatenate((cand[np.sort(sel)], np.full((2, order), -1, np.int32)))
    sf = 5.0
    scene = G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads))
    tracer = G.ExhaustivePathTracer(smoothing_factor=sf, batch_size=batch_size)
    got = scene.trace_paths(path_candidates=cand, solver=tracer)
    assert tuple(got.mask.shape) == (3, 5, cand.shape[0])
    full, m = _ref_trace(V, Tr, tx, rx, cand, mask, assume_quads, sf, batch_size)
    _assert_masks_close(_np(got.mask), m.numpy())
    np.testing.assert_allclose(_np(got.vertices), full.numpy(), rtol=1e-5, atol=1e-5)
    assert (_np(got.mask)[..., -2:] == 0).all() and (_np(got.vertices)[..., -2:, :, :] == 0).all()
    mm = m.numpy()
    mm = mm[~np.isnan(mm)]
    assert ((mm > 1e-3) & (mm < 1 - 1e-3)).sum() >= 5  # the comparison is not 0 == 0 only
    if not assume_quads:
        assert (_np(got.objects)[..., -1, 1:-1] == -1).all()


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_trace_smooth_vjp(G, order, assume_quads):
    """d(sum w*mask + sum wv*vertices)/d(tx, rx, mesh vertices): HIP reverse vs float64 autograd.
    Small perturbed box so that every term of the min (inside, blocked, too-small) is exercised."""
    rng = np.random.default_rng(3 + order)
    V, Tr = orc.box_mesh(4.0, 3.0, 2.5, with_top=True)
    V = (V + rng.normal(size=V.shape) * 0.05).astype(np.float32)
    n = 6 if assume_quads else 12
    cand = orc.generate_all_path_candidates(n, order).astype(np.int64) * (2 if assume_quads else 1)
    if cand.shape[0] > 200:
        cand = cand[np.sort(rng.choice(cand.shape[0], 200, replace=False))]
    mask = np.ones(12, bool)
    mask[10:] = False  # the top is inactive: candidates through it are multiplied by 0
    tx = np.array([[0.7, -0.4, 0.3], [-1.1, 0.2, -0.6]], np.float32)
    rx = np.array([[-0.9, 0.5, -0.2], [0.2, 0.9, 0.6], [0.70001, -0.4, 0.3]], np.float32)  # rx[2] ~ tx[0]
    w = rng.normal(size=(2, 3, cand.shape[0]))
    wv = rng.normal(size=(2, 3, cand.shape[0], order + 2, 3)) * 1e-2
    sf = 3.0

    def ref(dtype):
        ins = [torch.tensor(a, dtype=dtype, requires_grad=True) for a in (V, tx, rx)]
        full, m = tr.trace_smooth(ins[0], torch.tensor(Tr, dtype=torch.long), ins[1], ins[2], torch.tensor(cand),
                                  mask=torch.tensor(mask), assume_quads=assume_quads, epsilon=EPS, hit_tol=TOL,
                                  min_len=MINLEN, smoothing_factor=sf, batch_size=5)
        ((m * torch.tensor(w, dtype=dtype)).sum() + (full * torch.tensor(wv, dtype=dtype)).sum()).backward()
        return m.detach().numpy(), [t.grad.numpy() for t in ins]

    m64, g64 = ref(torch.float64)
    _, g32 = ref(torch.float32)
    vg, txg, rxg = (torch.tensor(a, device="cuda", requires_grad=True) for a in (V, tx, rx))
    scene = G.Scene(txg, rxg, G.Mesh(vg, Tr, mask=mask, assume_quads=assume_quads))
    got = scene.trace_paths(path_candidates=cand.astype(np.int32),
                            solver=G.ExhaustivePathTracer(smoothing_factor=sf, batch_size=5))
    np.testing.assert_allclose(_np(got.mask), m64, rtol=1e-4, atol=1e-5)
    ((got.mask * torch.tensor(w, dtype=torch.float32, device="cuda")).sum()
     + (got.vertices * torch.tensor(wv, dtype=torch.float32, device="cuda")).sum()).backward()
    _check_grads([_np(vg.grad), _np(txg.grad), _np(rxg.grad)], g32, g64, ("vertices", "tx", "rx"))


def test_trace_smooth_mask_only_and_vertices_only_cotangents(G, goldens, two_buildings):
    """Either cotangent alone reaches the reverse kernel; vertices-only == the hard-mode VJP."""
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    outs = []
    for sf in (2.0, None):
        vg = torch.tensor(V, device="cuda", requires_grad=True)
        txg = torch.tensor(np.asarray(g["tx"], np.float32), device="cuda", requires_grad=True)
        scene = G.Scene(txg, np.asarray(g["rx"], np.float32), G.Mesh(vg, Tr))
        p = scene.trace_paths(1, solver=G.ExhaustivePathTracer(smoothing_factor=sf))
        p.vertices.square().sum().backward()
        outs.append((_np(vg.grad), _np(txg.grad)))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-4)
    vg = torch.tensor(V, device="cuda", requires_grad=True)
    scene = G.Scene(np.asarray(g["tx"], np.float32), np.asarray(g["rx"], np.float32), G.Mesh(vg, Tr))
    scene.trace_paths(1, solver=G.ExhaustivePathTracer(smoothing_factor=20.0)).mask.sum().backward()
    assert torch.isfinite(vg.grad).all() and float(vg.grad.abs().max()) > 0


def test_trace_smooth_nonfinite_paths(G, goldens, two_buildings):
    """Order 2 on the two-buildings scene contains candidates whose image path is infinite: NaN
    confidence like the reference (jnp.min propagates NaN), zeroed vertices, and a finite (zero)
    contribution to the gradients."""
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    vg = torch.tensor(V, device="cuda", requires_grad=True)
    scene = G.Scene(np.asarray(g["tx"], np.float32), np.asarray(g["rx"], np.float32), G.Mesh(vg, Tr))
    p = scene.trace_paths(2, solver=G.ExhaustivePathTracer(smoothing_factor=0.5))
    nan = torch.isnan(p.mask)
    assert bool(nan.any()) and bool((p.vertices[nan] == 0).all())
    assert not bool((p.mask[~nan] < 0).any())
    torch.nan_to_num(p.mask).sum().backward()
    assert torch.isfinite(vg.grad).all()


def test_trace_smooth_interfaces(G, goldens, two_buildings):
    g = goldens["advanced_path_tracing_example"]
    scene = G.Scene(np.asarray(g["tx"], np.float32), np.asarray(g["rx"], np.float32),
                    G.Mesh(two_buildings["vertices"], two_buildings["triangles"]))
    tracer = G.ExhaustivePathTracer(smoothing_factor=50.0, confidence_threshold=0.9)
    chunks = list(scene.trace_paths(1, solver=G.ExhaustivePathTracer(smoothing_factor=50.0, chunk_size=7)))
    whole = scene.trace_paths(1, solver=tracer)
    assert whole.confidence_threshold == 0.9
    np.testing.assert_array_equal(np.concatenate([_np(c.mask) for c in chunks]), _np(whole.mask))
    with pytest.raises(NotImplementedError):
        scene.trace_paths(1, solver=tracer, compact=True)
    with pytest.warns(UserWarning, match="smoothing' is currently ignored"):
        hyb = scene.trace_paths(1, solver=G.HybridPathTracer(smoothing_factor=50.0, num_rays=100_000))
    assert hyb.mask.dtype == torch.float32  # warned, and forwarded all the same (SV:1173)
    empty = G.Scene(g["tx"], g["rx"], G.Mesh.empty()).trace_paths(0, solver=tracer)
    assert empty.mask.dtype == torch.float32 and _np(empty.mask).tolist() == [1.0]
