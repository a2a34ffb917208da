"""Visibility by ray launching + hybrid tracer ("next" row f2).

CPU part: the NumPy restatement (oracle) against the reference's known answers
(differt/tests/geometry/test_utils.py:717-827).  GPU part: HIP kernels vs the oracle; hybrid tracer
== exhaustive tracer on the two-buildings goldens (differt/tests/geometry/test_scene.py:116-260,
`method="hybrid"` rows) and never reports a path the exhaustive tracer does not.
"""

from __future__ import annotations

import numpy as np
import pytest

import oracle as orc


@pytest.fixture(scope="module")
def cube_tv():
    V, Tr = orc.box_mesh(with_top=True)
    return orc.triangle_vertices(V, Tr)


@pytest.mark.parametrize("vertex,expected", [([2.0, 0.0, 0.0], 2), ([2.0, 2.0, 0.0], 4), ([2.0, 2.0, 2.0], 6)])
@pytest.mark.parametrize("num_rays", [20, 10_000, 1_000_000, 1])
def test_oracle_cube_counts(cube_tv, vertex, expected, num_rays):
    """test_utils.py:717-767: a cube seen from a face / an edge / a corner direction."""
    if num_rays == 1_000_000 and expected != 6:
        pytest.skip("1e6 rays x 12 triangles on the CPU oracle: one case is enough")
    vis = orc.triangles_visible_from_vertex(np.asarray(vertex, np.float32), cube_tv, num_rays=num_rays)
    if num_rays == 1:
        assert vis.sum() != expected  # "Impossible to find all visible faces with few rays"
    else:
        assert vis.sum() == expected


def _box_in_box():
    Vo, To = orc.box_mesh(4.0, 4.0, 4.0)
    Vi, Ti = orc.box_mesh(1.0, 1.0, 1.0)
    V, Tr = np.concatenate((Vo, Vi)), np.concatenate((To, Ti + 8))
    mask = np.concatenate((np.ones(10, bool), np.zeros(10, bool)))
    return V, Tr, mask


def test_oracle_inside_masked_box():
    """test_utils.py:770-827.  The unmasked RX count (12 in the reference) hinges on lattice rays
    grazing an edge exactly, i.e. on the last ulp of XLA's cos/arccos: NumPy's give 11 -- the one
    known answer of this row that the oracle does not reproduce (documented in DESIGN.md)."""
    V, Tr, mask = _box_in_box()
    tv = orc.triangle_vertices(V, Tr)
    tx, rx = np.asarray([-1.0, 0, 0], np.float32), np.asarray([1.0, 0, 0], np.float32)
    a = orc.triangles_visible_from_vertex(tx, tv, mask, num_rays=200_000)
    b = orc.triangles_visible_from_vertex(rx, tv, mask, num_rays=200_000)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, mask)
    au = orc.triangles_visible_from_vertex(tx, tv, None, num_rays=200_000)
    bu = orc.triangles_visible_from_vertex(rx, tv, None, num_rays=200_000)
    assert (au != bu).any() and au.sum() == 11 and bu.sum() in (11, 12)


def test_oracle_lattice_properties():
    """_utils.py:369-490: unit vectors, inside the frustum; ValueError for n <= 0."""
    xyz = orc.fibonacci_lattice(1000)
    np.testing.assert_allclose(np.linalg.norm(xyz, axis=-1), 1.0, atol=1e-6)
    fr = np.array([[1.0, 0.5, -0.3], [2.0, 1.2, 0.9]], np.float32)
    rpa = orc.cartesian_to_spherical(orc.fibonacci_lattice(5000, frustum=fr))
    assert (rpa[:, 1] >= 0.5 - 1e-5).all() and (rpa[:, 1] <= 1.2 + 1e-5).all()
    assert (rpa[:, 2] >= -0.3 - 1e-5).all() and (rpa[:, 2] <= 0.9 + 1e-5).all()
    with pytest.raises(ValueError):
        orc.fibonacci_lattice(0)
    # the split-modulus trick keeps azimuths distinct for large i (:426-462)
    big = orc.cartesian_to_spherical(orc.fibonacci_lattice(12_000_000)[-20000:])
    assert len(np.unique(np.round(big[:, 2], 5))) > 5000


def _frustum_known_answers():
    """The four hand-made cases of differt/tests/geometry/test_utils.py:297-378:
    (viewer, world vertices, check(a_min, a_max))."""
    a3 = np.array([0.6, 0.7, 0.8])
    a1, a2 = np.pi - 0.15, -np.pi + 0.15
    a8 = np.linspace(-np.pi, np.pi, 8, endpoint=False)
    ring = lambda a: np.stack([np.cos(a), np.sin(a), np.zeros_like(a)], axis=-1)  # noqa: E731
    corridor = [[0, 2, 0], [10, 2, 0], [0, 2, 3], [10, 2, 3], [0, -2, 0], [10, -2, 0], [0, -2, 3], [10, -2, 3]]
    return [
        ([0, 0, 0], ring(a3), lambda lo, hi: hi - lo < np.pi and abs(lo - 0.6) < 0.05 and abs(hi - 0.8) < 0.05),
        ([0, 0, 0], ring(np.array([a1, a2])), lambda lo, hi: hi - lo < np.pi),  # straddles +-pi: [0, 2 pi) domain
        ([0, 0, 0], ring(a8), lambda lo, hi: abs(lo + np.pi) < 1e-5 and abs(hi - np.pi) < 1e-5),  # surrounded
        ([5, 0, 1.5], np.array(corridor, float), lambda lo, hi: hi - lo >= 1.9 * np.pi),
    ]


def test_oracle_viewing_frustum_known_answers():
    for viewer, verts, check in _frustum_known_answers():
        fr = orc.viewing_frustum(np.asarray(viewer, np.float32), np.asarray(verts, np.float32))
        assert fr.shape == (2, 3) and check(float(fr[0, 2]), float(fr[1, 2])), fr


# ---------------------------------------------------------------------------- GPU ----
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


@gpu
def test_gpu_viewing_frustum_known_answers(G):
    """test_utils.py:297-378 through drt_viewing_frustum_points (frustum of arbitrary world points)."""
    import torch

    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream

    for viewer, verts, check in _frustum_known_answers():
        v = torch.tensor(np.asarray(viewer, np.float32)[None], device="cuda")
        w = torch.tensor(np.asarray(verts, np.float32), device="cuda").contiguous()
        out = torch.empty((1, 2, 3), dtype=torch.float32, device="cuda")
        _lib.call("drt_viewing_frustum_points", ptr(v), 1, ptr(w), w.shape[0], ptr(out), stream())
        got = _np(out)[0]
        assert check(float(got[0, 2]), float(got[1, 2])), got
        np.testing.assert_allclose(got, orc.viewing_frustum(np.asarray(viewer, np.float32), np.asarray(verts, np.float32)),
                                   atol=5e-6)


@gpu
def test_gpu_fibonacci_lattice_properties(G):
    """differt/tests/geometry/test_utils.py:381-436: unit length, both hemispheres, frustum bounds, and no
    azimuth "hatching" in the tail of a 1e7-point lattice (the split-modulus trick of :426-462)."""
    for n in (1, 10, 1000):
        pts = _np(G.fibonacci_lattice(n))
        assert pts.shape == (n, 3)
        np.testing.assert_allclose(np.linalg.norm(pts, axis=-1), 1.0, atol=1e-5)
    pts = _np(G.fibonacci_lattice(1000))
    assert (pts[:, 2] > 0).any() and (pts[:, 2] < 0).any()
    fr = np.array([[0.0, np.pi / 4, 0.1], [1.0, np.pi / 2, 0.9]], np.float32)
    sph = _np(G.cartesian_to_spherical(G.fibonacci_lattice(500, frustum=fr)))
    assert sph.shape == (500, 3)
    assert (sph[:, 1] >= np.pi / 4 - 1e-4).all() and (sph[:, 1] <= np.pi / 2 + 1e-4).all()
    assert (sph[:, 2] >= 0.1 - 1e-4).all() and (sph[:, 2] <= 0.9 + 1e-4).all()
    n = 10_000_000
    i = np.arange(n - 10_000, n, dtype=np.float32)
    assert len(np.unique((i * np.float32(0.6180339887498949)) % np.float32(1.0))) < 1000  # the naive form collapses
    tail = np.round(_np(G.fibonacci_lattice(n)[-10_000:]), 4)
    assert len(np.unique(tail, axis=0)) > 5000


@gpu
def test_gpu_sample_triangles_visibility(G):
    """Extension: interior sample points of every face complement the lattice rays.  On a small scene the
    sample pass must equal a NumPy restatement (segment viewpoint -> sample blocked by another active
    triangle before t = 1 - 1e-4, Moller-Trumbore of the oracle); it only ever adds faces."""
    rng = np.random.default_rng(12)
    outer, inner = orc.box_mesh(8.0, 6.0, 5.0, with_top=True), orc.box_mesh(2.0, 2.0, 2.0, with_top=True)
    V = np.concatenate((outer[0], inner[0] + np.float32([1.0, 0.5, -0.3])))
    Tr = np.concatenate((outer[1], inner[1] + 8)).astype(np.int32)
    mask = np.ones(len(Tr), bool)
    mask[5] = False
    views = rng.uniform(-3.5, 3.5, (6, 3)).astype(np.float32) * np.float32([1, 0.8, 0.6])
    tv = orc.triangle_vertices(V, Tr)
    w = np.array([[1 / 3, 1 / 3, 1 / 3], [0.8, 0.1, 0.1], [0.1, 0.8, 0.1], [0.1, 0.1, 0.8], [0.45, 0.45, 0.1],
                  [0.1, 0.45, 0.45], [0.45, 0.1, 0.45]], np.float32)
    for m in (None, mask):
        mesh = G.Mesh(V, Tr, mask=m)
        lat = _np(mesh.triangles_visible_from_vertex(views, num_rays=1, accel="bvh"))
        got = _np(mesh.triangles_visible_from_vertex(views, num_rays=1, accel="bvh", sample_triangles=True))
        assert (got | lat == got).all()  # only adds
        exp = lat.copy()
        act = np.ones(len(Tr), bool) if m is None else m
        for b, o in enumerate(views):
            for j in np.flatnonzero(act):
                pts = ((tv[j, 0][None] * w[:, :1]).astype(np.float32) + (tv[j, 1][None] * w[:, 1:2]).astype(np.float32)).astype(np.float32)
                pts = (pts + (tv[j, 2][None] * w[:, 2:3]).astype(np.float32)).astype(np.float32)
                others = act.copy()
                others[j] = False
                t, hit = orc.ray_intersect_triangle(o[None, None, :], (pts - o)[:, None, :], tv[None])
                blocked = ((t < np.float32(1.0 - 1e-4)) & hit & others[None, :]).any(axis=1)
                exp[b, j] |= bool((~blocked).any())
        np.testing.assert_array_equal(got, exp)
        assert got.sum() > lat.sum() + 10
    dense = _np(G.Mesh(V, Tr).triangles_visible_from_vertex(views, num_rays=2_000_000, accel="bvh"))
    both = _np(G.Mesh(V, Tr).triangles_visible_from_vertex(views, num_rays=1000, accel="bvh", sample_triangles=True))
    assert (dense & ~both).sum() <= 2  # the sample pass recovers (nearly) everything dense lattice sampling sees
    with pytest.raises(ValueError, match="sample_triangles needs accel"):
        G.Mesh(V, Tr).triangles_visible_from_vertex(views, num_rays=10, sample_triangles=True)


@gpu
def test_gpu_spherical_conversions(G):
    """geometry/_utils.py:930-993 (round trip and the zero vector), vs the oracle."""
    rng = np.random.default_rng(4)
    xyz = (rng.normal(size=(5, 300, 3)) * 10).astype(np.float32)
    xyz[0, 0] = 0
    xyz[0, 1] = [0, 0, 2.5]
    rpa = G.cartesian_to_spherical(xyz)
    np.testing.assert_allclose(_np(rpa), orc.cartesian_to_spherical(xyz), rtol=1e-6, atol=1e-6)
    assert _np(rpa)[0, 0].tolist() == [1.0, np.float32(np.pi / 2), 0.0]
    back = G.spherical_to_cartesian(rpa)
    np.testing.assert_allclose(_np(back)[0, 1:], xyz[0, 1:], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(_np(back), orc.spherical_to_cartesian(_np(rpa)), rtol=1e-6, atol=1e-6)
    unit = G.spherical_to_cartesian(_np(rpa)[..., 1:])
    assert tuple(unit.shape) == (5, 300, 3)
    np.testing.assert_allclose(np.linalg.norm(_np(unit), axis=-1), 1.0, atol=1e-6)


@gpu
def test_gpu_lattice_and_frustum_vs_oracle(G, cube_tv):
    rng = np.random.default_rng(0)
    for n in (1, 7, 1000, 300_000):
        np.testing.assert_allclose(_np(G.fibonacci_lattice(n)), orc.fibonacci_lattice(n), atol=2e-6)
    views = rng.uniform(-5, 5, (6, 3)).astype(np.float32)
    tv = rng.normal(size=(40, 3, 3)).astype(np.float32) * 2
    act = rng.random(40) > 0.3
    world = np.concatenate((tv, tv.mean(axis=-2, keepdims=True, dtype=np.float32)), axis=-2).reshape(-1, 3)
    for a in (None, act):
        exp = orc.viewing_frustum(views, world, active_vertices=None if a is None else np.repeat(a, 4))
        got = _np(G.viewing_frustum(views, world, active_vertices=None if a is None else np.repeat(a, 4)))
        np.testing.assert_allclose(got, exp, atol=5e-6)
        # reduce=True (_utils.py:838-846 with axis=None): one frustum over every viewer; per-viewer point
        # sets and masks broadcast like in the reference
        per_view = np.stack([world + 0.01 * i for i in range(len(views))]).astype(np.float32)
        r1 = _np(G.viewing_frustum(views, per_view, reduce=True))
        assert r1.shape == (2, 3)
        each = _np(G.viewing_frustum(views, per_view))
        assert each.shape == (len(views), 2, 3)
        np.testing.assert_allclose(r1[0, 0], each[:, 0, 0].min(), rtol=1e-6)
        np.testing.assert_allclose(r1[1, 0], each[:, 1, 0].max(), rtol=1e-6)
        np.testing.assert_allclose(r1[0, 1], each[:, 0, 1].min(), rtol=1e-6)
        np.testing.assert_allclose(r1[1, 1], each[:, 1, 1].max(), rtol=1e-6)
        for b in range(2):
            e = orc.fibonacci_lattice(5000, frustum=exp[b])
            g = _np(G.fibonacci_lattice(5000, frustum=got[b]))
            np.testing.assert_allclose(g, e, atol=2e-5)
    with pytest.raises(ValueError):
        G.fibonacci_lattice(0)


@gpu
@pytest.mark.parametrize("vertex,expected", [([2.0, 0.0, 0.0], 2), ([2.0, 2.0, 0.0], 4), ([2.0, 2.0, 2.0], 6)])
@pytest.mark.parametrize("num_rays", [20, 10_000, 1_000_000])
def test_gpu_cube_counts(G, cube_tv, vertex, expected, num_rays):
    """test_utils.py:717-767 on the HIP kernels, and equality with the oracle's visible set."""
    vis = _np(G.triangles_visible_from_vertex(np.asarray(vertex, np.float32), cube_tv, num_rays=num_rays))
    assert vis.sum() == expected
    if num_rays <= 10_000:
        np.testing.assert_array_equal(vis, orc.triangles_visible_from_vertex(np.asarray(vertex, np.float32), cube_tv, num_rays=num_rays))


@gpu
def test_gpu_inside_masked_box_and_mesh_method(G):
    """test_utils.py:770-827 + test_mesh.py:2075-2090 (Mesh method == free function)."""
    V, Tr, mask = _box_in_box()
    mesh = G.Mesh(V, Tr, mask=mask)
    pts = np.asarray([[-1.0, 0, 0], [1.0, 0, 0]], np.float32)
    vis = _np(mesh.triangles_visible_from_vertex(pts))
    assert vis.shape == (2, 20)
    np.testing.assert_array_equal(vis[0], mask)
    np.testing.assert_array_equal(vis[1], mask)
    free = _np(G.triangles_visible_from_vertex(pts, orc.triangle_vertices(V, Tr), mask))
    np.testing.assert_array_equal(free, vis)
    un = _np(G.triangles_visible_from_vertex(pts, orc.triangle_vertices(V, Tr), None))
    assert (un[0] != un[1]).any() and un[0].sum() in (11, 12) and un[1].sum() in (11, 12)
    assert _np(G.Mesh.empty().triangles_visible_from_vertex(pts)).shape == (2, 0)


@gpu
def test_gpu_visibility_random_scene_vs_oracle(G):
    """Random boxes: the GPU set must equal the oracle set up to triangles grazed by a single ray
    (ulp-different lattice directions), and never contain a triangle the oracle does not see with 4x
    the rays."""
    from conftest import canyon_scene

    rng = np.random.default_rng(5)
    V, Tr = canyon_scene(rng, nextra=6)
    tv = orc.triangle_vertices(V, Tr)
    view = np.asarray([3.0, 0.5, 6.0], np.float32)
    got = _np(G.triangles_visible_from_vertex(view, tv, num_rays=100_000))
    exp = orc.triangles_visible_from_vertex(view, tv, num_rays=100_000)
    dense = orc.triangles_visible_from_vertex(view, tv, num_rays=400_000)
    assert (got != exp).sum() <= 2
    assert not (got & ~(dense | exp)).any()
    assert got.sum() >= 20


@gpu
@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_gpu_hybrid_equals_exhaustive_on_goldens(G, goldens, two_buildings, order, assume_quads):
    """test_scene.py:116-260 with solver="hybrid" (rtol 1e-6) and test_scene.py:726-759
    (hybrid == exhaustive)."""
    g = goldens["advanced_path_tracing_example"]
    exp = g["orders"][str(order)]
    scene = G.Scene(np.asarray(g["tx"], np.float32), np.asarray(g["rx"], np.float32),
                    G.Mesh(two_buildings["vertices"], two_buildings["triangles"], assume_quads=assume_quads))
    hyb = scene.trace_paths(order, solver="hybrid", num_rays=200_000)
    exh = scene.trace_paths(order)
    eo = np.asarray(exp["objects"], np.int32)
    if assume_quads:
        eo = eo - eo % 2
    np.testing.assert_array_equal(_np(hyb.masked_objects), eo)
    np.testing.assert_array_equal(_np(hyb.masked_objects), _np(exh.masked_objects))
    np.testing.assert_array_equal(_np(hyb.masked_vertices), _np(exh.masked_vertices))
    ev = orc.assemble_path(np.asarray(g["tx"], np.float32),
                           np.asarray(exp["path_vertices"], np.float32).reshape(1, order, 3),
                           np.asarray(g["rx"], np.float32))
    np.testing.assert_allclose(_np(hyb.masked_vertices), ev, rtol=g["rtol"])
    if order >= 1:
        assert hyb.mask.shape[-1] < exh.mask.shape[-1]  # the candidate set really is pruned
    cp = scene.trace_paths(order, solver="hybrid", num_rays=200_000, compact=True)
    np.testing.assert_array_equal(_np(cp.objects), eo)


@gpu
def test_gpu_visibility_batched_triangle_sets(G, cube_tv):
    """_utils.py:1540-1772 accepts `*#batch` on the triangle set and the mask too: each batch entry sees
    its own triangles (here: the cube and a shifted copy; second entry with two faces masked)."""
    import torch

    tv = np.stack((cube_tv, cube_tv + np.float32(0.25)))  # [2, 12, 3, 3]
    act = np.ones((2, 12), bool)
    act[1, :4] = False
    vertex = np.array([[2.0, 2.0, 2.0], [2.0, 2.0, 2.0]], np.float32)
    got = G.triangles_visible_from_vertex(vertex, tv, act, num_rays=100_000)
    assert tuple(got.shape) == (2, 12)
    for b in range(2):
        one = G.triangles_visible_from_vertex(vertex[b], tv[b], act[b], num_rays=100_000)
        assert torch.equal(got[b], one)
    assert int(got[0].sum()) == 6 and not bool(got[1, :4].any())
