"""Pin the CPU oracle against the reference's own golden vectors (tests/golden/).

Each test names the reference test it mirrors.  These run on CPU (`-m "not gpu"`).
"""

from __future__ import annotations

import numpy as np
import pytest

import oracle as orc


# ---------------------------------------------------------------- enumerator ----
def test_generate_all_path_candidates_tables(goldens):
    """differt/tests/geometry/test_utils.py:448-490; ordered as graph.rs:1488-1513."""
    for case in goldens["generate_all_path_candidates"]["cases"]:
        got = orc.generate_all_path_candidates(case["num_primitives"], case["order"])
        assert list(got.shape) == case["shape"], case
        expected = np.asarray(case["rows"], dtype=np.int64).reshape(case["shape"])
        np.testing.assert_array_equal(got, expected)  # already lexicographic, no sort needed


@pytest.mark.parametrize("n,order", [(3, 1), (3, 2), (3, 3), (5, 4), (7, 3), (12, 2)])
def test_candidates_count_and_order(n, order):
    """differt-core/tests/geometry/test_graph.py:158-205 count formula; graph.rs:1536-1549 sortedness."""
    got = orc.generate_all_path_candidates(n, order)
    assert got.shape == (n * (n - 1) ** (order - 1), order)
    assert (got[:, 1:] != got[:, :-1]).all()
    keys = [tuple(r) for r in got]
    assert keys == sorted(keys)
    assert len(set(keys)) == len(keys)


def test_complete_graph_equals_digraph():
    """graph.rs:1563-1577: CompleteGraph and DiGraph iterators agree."""
    for n, depth in [(4, 3), (5, 4), (6, 5), (3, 2)]:
        cg = orc.CompleteGraphIter(n, n, n + 1, depth, True).collect_array()
        g = orc.DiGraph.from_complete_graph(n)
        f, t = g.insert_from_and_to_nodes(direct_path=True)
        assert (f, t) == (n, n + 1)
        dg = g.all_paths_array(f, t, depth, include_from_and_to=True)
        np.testing.assert_array_equal(cg, dg)


def test_digraph_mask_equals_smaller_complete_graph():
    """differt-core/tests/geometry/test_graph.py:120-142 / SV:820-827."""
    n, order = 7, 3
    mask = np.array([1, 0, 1, 1, 0, 1, 1], dtype=bool)
    g = orc.DiGraph.from_complete_graph(n)
    f, t = g.insert_from_and_to_nodes()
    g.filter_by_mask(mask, fast_mode=True)
    got = g.all_paths_array(f, t, order + 2, include_from_and_to=False)
    active = np.flatnonzero(mask)
    expected = active[orc.generate_all_path_candidates(len(active), order)]
    np.testing.assert_array_equal(got, expected)


def test_complete_graph_in_graph_endpoints():
    """graph.rs:326-366: counts when from/to are graph nodes (brute force cross-check)."""
    import itertools

    for n, depth, fr, to in [(4, 4, 0, 3), (4, 4, 1, 1), (5, 3, 2, 7), (5, 5, 9, 1), (3, 2, 0, 0)]:
        it = orc.CompleteGraphIter(n, fr, to, depth, True)
        declared = len(it)
        got = it.collect_array()
        brute = []
        for mid in itertools.product(range(n), repeat=max(depth - 2, 0)):
            p = (fr, *mid, to)
            if all(a != b for a, b in zip(p[:-1], p[1:])):
                brute.append(p)
        if depth == 2 and fr == to:
            brute = []
        assert declared == len(brute)
        np.testing.assert_array_equal(got, np.asarray(brute, dtype=np.int64).reshape(len(brute), depth))


def test_complete_graph_overflow_flag():
    """graph.rs:368-375: overflow -> usize::MAX + warning."""
    it = orc.CompleteGraphIter(10**6, 10**6, 10**6 + 1, 8, False)
    assert it.overflowed and it.remaining == 2**64 - 1


# ---------------------------------------------------------------- Moller-Trumbore ----
def test_ray_intersect_triangle_known_hits(goldens):
    """differt/tests/geometry/test_utils.py:555-577."""
    g = goldens["ray_intersect_triangle_hits"]
    tri = np.asarray([g["triangle"]], dtype=np.float32)
    for case in g["cases"]:
        o = np.asarray(case["orig"], dtype=np.float32)
        d = np.asarray(case["dest"], dtype=np.float32) - o
        t, hit = orc.ray_intersect_triangle(o, d, tri)
        assert bool(((t < 1.0) & hit)[0]) == case["expected"]


def test_ray_intersect_triangle_t_and_hit(goldens):
    """differt/tests/geometry/test_utils.py:580-606 (exact equality)."""
    g = goldens["ray_intersect_triangle_t_and_hit"]
    o = np.asarray(g["ray_origin"], dtype=np.float32)
    d = np.asarray(g["ray_directions"], dtype=np.float32)
    tv = np.asarray(g["triangle_vertices"], dtype=np.float32)
    t, hit = orc.ray_intersect_triangle(o[None, None, :], d[:, None, :], tv)
    np.testing.assert_array_equal(t, np.asarray(g["expected_t"], dtype=np.float32))
    np.testing.assert_array_equal(hit, np.asarray(g["expected_hit"]))
    t2, hit2 = orc.ray_intersect_triangle_dense(np.broadcast_to(o, d.shape), d, tv)
    np.testing.assert_array_equal(t2, t)
    np.testing.assert_array_equal(hit2, hit)


def test_ray_intersect_triangle_hit_implies_positive_t(rng):
    """differt/tests/geometry/test_utils.py:609-629."""
    o = rng.normal(size=(15, 5, 3)).astype(np.float32)
    d = rng.normal(size=(15, 5, 3)).astype(np.float32)
    tv = rng.normal(size=(5, 3, 3)).astype(np.float32)
    t, hit = orc.ray_intersect_triangle(o, d, tv)
    assert t.shape == (15, 5) and (t[hit] > 0).all()


@pytest.mark.parametrize("epsilon", [None, 1e-6, 1e-2])
@pytest.mark.parametrize("hit_tol", [None, 0.0, 0.001, -0.5, 0.5])
@pytest.mark.parametrize("with_active", [True, False])
@pytest.mark.parametrize(
    "shapes",
    [((20, 10, 3), (20, 10, 3), (20, 10, 5, 3, 3)), ((10, 3), (10, 3), (1, 3, 3)), ((3,), (3,), (1, 3, 3))],
)
def test_ray_intersect_any_triangle_vs_dense(rng, shapes, epsilon, hit_tol, with_active):
    """differt/tests/geometry/test_utils.py:649-714 (property form, batch_size=11)."""
    so, sd, st = shapes
    o = rng.normal(size=so).astype(np.float32)
    d = rng.normal(size=sd).astype(np.float32)
    tv = rng.normal(size=st).astype(np.float32)
    act = np.ones(st[:-2], dtype=bool) if with_active else None
    tol = orc.DEFAULT_HIT_TOL if hit_tol is None else hit_tol
    got = orc.ray_intersect_any_triangle(o, d, tv, act, epsilon=epsilon, hit_tol=hit_tol, batch_size=11)
    et, eh = orc.ray_intersect_triangle(o[..., None, :], d[..., None, :], tv, epsilon=epsilon)
    expected = ((et < np.float32(1.0) - np.float32(tol)) & eh).any(axis=-1)
    np.testing.assert_array_equal(got, expected)


@pytest.mark.parametrize("epsilon", [None, 1e-2])
@pytest.mark.parametrize("with_active", [True, False])
@pytest.mark.parametrize(
    "shapes",
    [((10, 3), (1, 3), (30, 3, 3)), ((100, 3), (100, 3), (1, 300, 3, 3)), ((4, 3), (4, 3), (0, 3, 3))],
)
def test_first_triangle_hit_by_ray_vs_dense(rng, shapes, epsilon, with_active):
    """differt/tests/geometry/test_utils.py:910-962 (t equality; indices checked with the
    reference's own tile tie-break UT:1865-1867, 1886 restated independently in NumPy)."""
    so, sd, st = shapes
    o = rng.normal(size=so).astype(np.float32)
    d = rng.normal(size=sd).astype(np.float32)
    tv = rng.normal(size=st).astype(np.float32)
    act = np.ones(st[:-2], dtype=bool) if with_active else None
    gi, gt = orc.first_triangle_hit_by_ray(o, d, tv, act, batch_size=11, epsilon=epsilon)
    T = st[-3]
    if T == 0:
        assert (gi == -1).all() and np.isinf(gt).all()
        return
    et, eh = orc.ray_intersect_triangle(o[..., None, :], d[..., None, :], tv, epsilon=epsilon)
    et = np.where(eh, et, np.inf).astype(np.float32)
    np.testing.assert_array_equal(gt, et.min(axis=-1))
    # tile semantics: lowest index inside a tile, later tile wins ties
    bs = min(11, T)
    flat_t = et.reshape(-1, T)
    exp_idx = np.full(flat_t.shape[0], -1, dtype=np.int32)
    for r, row in enumerate(flat_t):
        best_t, best_i = np.inf, -1
        starts = list(range(0, (T // bs) * bs, bs))
        tiles = [(s, s + bs) for s in starts] + ([(T - T % bs, T)] if T % bs else [])
        for s, e in tiles:
            j = int(np.argmin(row[s:e]))
            tt = row[s + j]
            if not (best_t < tt):
                best_t, best_i = tt, (s + j if np.isfinite(tt) else s - 1)
        exp_idx[r] = best_i if np.isfinite(best_t) else -1
    np.testing.assert_array_equal(gi.reshape(-1), exp_idx)


def test_first_hit_tie_break_later_tile_wins():
    """UT:1865-1867 + UT:1886: duplicate triangles -> lowest index in tile, later tile wins."""
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    tv = np.stack([tri] * 6)
    o = np.array([[0.25, 0.25, 1.0]], dtype=np.float32)
    d = np.array([[0.0, 0.0, -1.0]], dtype=np.float32)
    idx, t = orc.first_triangle_hit_by_ray(o, d, tv, batch_size=4)
    assert t[0] == 1.0 and idx[0] == 4  # tiles [0..3], remainder [4,5]: later tile, lowest index
    idx, _ = orc.first_triangle_hit_by_ray(o, d, tv, batch_size=6)
    assert idx[0] == 0
    idx, _ = orc.first_triangle_hit_by_ray(o, d, tv, batch_size=2)
    assert idx[0] == 4


def test_empty_mesh_conventions():
    """UT:1441-1450, 1848-1857."""
    o = np.zeros((4, 3), np.float32)
    d = np.ones((4, 3), np.float32)
    tv = np.zeros((0, 3, 3), np.float32)
    assert not orc.ray_intersect_any_triangle(o, d, tv).any()
    i, t = orc.first_triangle_hit_by_ray(o, d, tv)
    assert (i == -1).all() and np.isinf(t).all()


# ---------------------------------------------------------------- image method ----
def test_image_of_vertex(goldens):
    """differt/tests/geometry/test_image_method.py:19-29."""
    g = goldens["image_of_vertex"]
    got = orc.image_of_vertex_with_respect_to_mirror(g["vertices"], g["mirror_vertices"], g["mirror_normals"])
    np.testing.assert_allclose(got, np.asarray(g["expected"], np.float32))


def test_intersection_of_ray_with_plane(goldens):
    """differt/tests/geometry/test_image_method.py:70-129."""
    g = goldens["intersection_of_ray_with_plane"]
    o = np.asarray(g["ray_origins"], np.float32)
    d = np.asarray(g["ray_end"], np.float32)[None, :] - o
    for case in g["cases"]:
        got = orc.intersection_of_ray_with_plane(o, d, [case["plane_vertex"]], [case["plane_normal"]])
        if case["expected"] == "inf":
            assert np.isposinf(got).all()
        elif case["expected"] == "origins":
            np.testing.assert_allclose(got, o)
        else:
            np.testing.assert_allclose(got, np.asarray(case["expected"], np.float32), atol=1e-7)


@pytest.mark.parametrize("batch", [(), (10,), (10, 20, 30)])
def test_image_method_corridor(goldens, rng, batch):
    """differt/tests/geometry/test_image_method.py:160-188 with the 'no-effect' noise of
    differt/tests/geometry/utils.py:28-54 (normal flips, in-plane mirror-vertex shifts)."""
    g = goldens["planar_mirrors_setup"]
    k = len(g["paths"])
    a = np.broadcast_to(np.asarray(g["from_vertex"], np.float32), (*batch, 3))
    b = np.broadcast_to(np.asarray(g["to_vertex"], np.float32), (*batch, 3))
    mv = np.broadcast_to(np.asarray(g["mirror_vertices"], np.float32), (*batch, k, 3)).copy()
    mn = np.broadcast_to(np.asarray(g["mirror_normals"], np.float32), (*batch, k, 3)).copy()
    exp = np.broadcast_to(np.asarray(g["paths"], np.float32), (*batch, k, 3))
    shift = rng.normal(size=mv.shape).astype(np.float32) * np.float32(10.0)
    shift = shift - (shift * mn).sum(-1, keepdims=True) * mn
    sign = rng.choice(np.array([1.0, -1.0], np.float32), size=mv.shape[:-1])
    got = orc.image_method(a, b, mv + shift, mn * sign[..., None])
    np.testing.assert_allclose(got, exp, atol=1e-5)
    # returned points lie on their planes (test_image_method.py:191-219)
    on_plane = ((got - (mv + shift)) * mn).sum(-1)
    np.testing.assert_allclose(on_plane, 0.0, atol=1e-4)
    ss = orc.consecutive_vertices_are_on_same_side_of_mirror(orc.assemble_path(a, got, b), mv + shift, mn)
    assert ss.shape == (*batch, k) and ss.all()


def test_same_side_requires_k_plus_2_vertices():
    """IM:422-424."""
    with pytest.raises(TypeError):
        orc.consecutive_vertices_are_on_same_side_of_mirror(
            np.zeros((3, 3), np.float32), np.zeros((2, 3), np.float32), np.zeros((2, 3), np.float32)
        )


def test_image_method_inf_propagation():
    """IM:123-135, 165-181: parallel mirror -> inf, propagated (never NaN)."""
    a = np.array([0.0, 0.0, 1.0], np.float32)
    b = np.array([1.0, 0.0, 1.0], np.float32)  # segment image->b parallel to the second mirror
    mv = np.array([[0, 0, 0], [0, 0, -1.0]], np.float32)
    mn = np.array([[1.0, 0, 0], [0, 0, 1.0]], np.float32)
    got = orc.image_method(a, b, mv, mn)
    assert not np.isnan(got).any()


# ---------------------------------------------------------------- mesh helpers ----
def test_box_tables(goldens):
    """ME:2172-2217."""
    v, t = orc.box_mesh(with_top=True)
    assert v.shape == (8, 3) and t.shape == (12, 3)
    np.testing.assert_array_equal(t, np.asarray(goldens["box_with_top"]["triangles"], np.int32))
    n = orc.mesh_normals(orc.triangle_vertices(v, t))
    np.testing.assert_allclose(np.linalg.norm(n, axis=-1), 1.0, rtol=1e-6)
    # each quad (2q, 2q+1) is planar: identical normals
    np.testing.assert_array_equal(n[0::2], n[1::2])


# ---------------------------------------------------------------- full pipeline ----
@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("mesh_mask", [False, True])
def test_trace_paths_two_buildings(goldens, two_buildings, order, assume_quads, mesh_mask):
    """differt/tests/geometry/test_scene.py:116-260 (exhaustive solver rows), rtol 1e-6."""
    g = goldens["advanced_path_tracing_example"]
    exp = g["orders"][str(order)]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    n_prim = Tr.shape[0] // 2 if assume_quads else Tr.shape[0]
    cand = orc.generate_all_path_candidates(n_prim, order).astype(np.int32)
    if assume_quads:
        cand = 2 * cand  # SV:842-843
    if order == 4 and not assume_quads:
        # 24*23^3 = 292008 candidates x 5 segments x 24 triangles: fine for the C oracle
        pass
    mask = np.ones(Tr.shape[0], dtype=bool) if mesh_mask else None
    out = orc.trace_path_candidates(
        V, Tr, g["tx"], g["rx"], cand, mask=mask, assume_quads=assume_quads
    )
    m = out["mask"].reshape(-1)
    mv = out["vertices"].reshape(-1, order + 2, 3)[m]
    mo = out["objects"].reshape(-1, order + 2)[m]
    exp_objects = np.asarray(exp["objects"], dtype=np.int32)
    if assume_quads:
        exp_objects = exp_objects - exp_objects % 2  # test_scene.py:183-184
    exp_inner = np.asarray(exp["path_vertices"], np.float32).reshape(1, order, 3)
    exp_vertices = orc.assemble_path(
        np.asarray(g["tx"], np.float32), exp_inner, np.asarray(g["rx"], np.float32)
    )
    np.testing.assert_array_equal(mo, exp_objects)
    np.testing.assert_allclose(mv, exp_vertices, rtol=g["rtol"])
    # law of reflection (test_scene.py:248-260)
    if order > 0:
        nrm = orc.mesh_normals(orc.triangle_vertices(V, Tr))[mo[:, 1:-1]]
        rays, _ = orc.normalize(np.diff(mv, axis=-2))
        di = (-rays[:, :-1] * nrm).sum(-1)
        dr = (rays[:, 1:] * nrm).sum(-1)
        np.testing.assert_allclose(di, dr, rtol=1e-4, atol=1e-6)


def test_trace_mask_equals_submesh(two_buildings, goldens):
    """differt/tests/geometry/test_scene.py:585-647: masking triangles == removing them."""
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    rng = np.random.default_rng(7)
    mask = rng.random(Tr.shape[0]) > 0.3
    mask[[8, 9, 22]] = True
    order = 2
    cand = orc.generate_all_path_candidates(Tr.shape[0], order).astype(np.int32)
    full = orc.trace_path_candidates(V, Tr, g["tx"], g["rx"], cand, mask=mask)
    keep = np.flatnonzero(mask)
    sub_cand = orc.generate_all_path_candidates(len(keep), order).astype(np.int32)
    sub = orc.trace_path_candidates(V, Tr[keep], g["tx"], g["rx"], sub_cand)
    fm, sm = full["mask"].reshape(-1), sub["mask"].reshape(-1)
    np.testing.assert_array_equal(
        full["objects"].reshape(-1, 4)[fm][:, 1:-1], keep[sub["objects"].reshape(-1, 4)[sm][:, 1:-1]]
    )
    np.testing.assert_array_equal(
        full["vertices"].reshape(-1, 4, 3)[fm], sub["vertices"].reshape(-1, 4, 3)[sm]
    )


def test_trace_empty_and_padding_rows(two_buildings, goldens):
    """SV:566-573 (no candidates) and SV:912-918 (padded -1 rows are invalid)."""
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    out = orc.trace_path_candidates(V, Tr, g["tx"], g["rx"], np.zeros((0, 2), np.int32))
    assert out["vertices"].shape == (1, 1, 0, 4, 3)
    out = orc.trace_path_candidates(V, Tr, g["tx"], g["rx"], np.array([[8], [-1]], np.int32))
    assert out["mask"].reshape(-1).tolist() == [True, False]
    assert (out["vertices"][0, 0, 1] == 0).all()


def test_config1_box_order1(goldens):
    """BASELINE.json configs[0]: 1 TX, 1 RX, 12-triangle box, order 1 (plumbing)."""
    V, Tr = orc.box_mesh(with_top=True)
    tx, rx = [0.1, -0.2, 0.05], [-0.3, 0.25, -0.1]
    cand = orc.generate_all_path_candidates(12, 1).astype(np.int32)
    out = orc.trace_path_candidates(V, Tr, tx, rx, cand)
    m = out["mask"].reshape(-1)
    # inside a closed box every face reflects exactly once; each quad = 2 triangles, the hit point
    # lies in exactly one of them (or on the shared diagonal -> both)
    assert 6 <= m.sum() <= 12
    q = orc.trace_path_candidates(V, Tr, tx, rx, 2 * orc.generate_all_path_candidates(6, 1).astype(np.int32), assume_quads=True)
    assert q["mask"].sum() == 6
