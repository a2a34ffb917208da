"""GPU parity of the image-method operators and the fused tracer (through the C ABI) vs the oracle.

Bar (BASELINE.json north-star): valid-path masks and object indices bit-exact; path vertices
<= 1e-5 rel (they are in fact bit-identical: same operation order, no FMA); gradients <= 1e-5 rel vs
torch.autograd over the torch restatement of the reference (oracle/torch_ref.py).
Mirrors differt/tests/geometry/test_image_method.py and test_scene.py:116-260, 334-364, 444-647.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc
from oracle import torch_ref

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------ image method ----
def test_image_operators_known_answers(G, goldens):
    """test_image_method.py:19-29, 70-129."""
    g = goldens["image_of_vertex"]
    got = G.image_of_vertex_with_respect_to_mirror(g["vertices"], g["mirror_vertices"], g["mirror_normals"])
    np.testing.assert_array_equal(_np(got), np.asarray(g["expected"], np.float32))
    g = goldens["intersection_of_ray_with_plane"]
    o = np.asarray(g["ray_origins"], np.float32)
    d = np.asarray(g["ray_end"], np.float32)[None] - o
    for case in g["cases"]:
        got = _np(G.intersection_of_ray_with_plane(o, d, [case["plane_vertex"]], [case["plane_normal"]]))
        exp = orc.intersection_of_ray_with_plane(o, d, [case["plane_vertex"]], [case["plane_normal"]])
        np.testing.assert_array_equal(_bits(got), _bits(exp))
        if case["expected"] == "inf":
            assert np.isposinf(got).all()


@pytest.mark.parametrize("batch", [(), (10,), (10, 20, 30)])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 8])
def test_image_method_random_bit_exact(G, rng, batch, k):
    """test_image_method.py:160-219 shapes; random mirrors, incl. inf propagation."""
    a = rng.normal(size=(*batch, 3)).astype(np.float32)
    b = rng.normal(size=(*batch, 3)).astype(np.float32)
    mv = rng.normal(size=(*batch, k, 3)).astype(np.float32)
    mn, _ = orc.normalize(rng.normal(size=(*batch, k, 3)).astype(np.float32))
    exp = orc.image_method(a, b, mv, mn)
    got = G.image_method(a, b, mv, mn)
    assert tuple(got.shape) == (*batch, k, 3)
    np.testing.assert_array_equal(_bits(_np(got)), _bits(exp))
    full = orc.assemble_path(a, exp, b)
    ss = G.consecutive_vertices_are_on_same_side_of_mirror(full, mv, mn)
    np.testing.assert_array_equal(_np(ss), orc.consecutive_vertices_are_on_same_side_of_mirror(full, mv, mn))


def test_image_method_corridor_and_empty(G, goldens):
    """fixtures.py:82-117 golden corridor; k == 0 -> empty (IM:349-358); TypeError (IM:422-424)."""
    g = goldens["planar_mirrors_setup"]
    got = G.image_method(g["from_vertex"], g["to_vertex"], g["mirror_vertices"], g["mirror_normals"])
    np.testing.assert_allclose(_np(got), np.asarray(g["paths"], np.float32), atol=1e-7)
    e = G.image_method(np.zeros((4, 3), np.float32), np.ones((4, 3), np.float32),
                       np.zeros((4, 0, 3), np.float32), np.zeros((4, 0, 3), np.float32))
    assert tuple(e.shape) == (4, 0, 3)
    with pytest.raises(TypeError):
        G.consecutive_vertices_are_on_same_side_of_mirror(np.zeros((3, 3)), np.zeros((2, 3)), np.zeros((2, 3)))


def test_image_method_parallel_inf_no_nan(G):
    """IM:123-135, 165-181 + test_image_method.py:109-117: inf outputs, NaN-free gradients."""
    a = np.array([[0.0, 0.0, 1.0]], np.float32)
    b = np.array([[1.0, 0.0, -1.0]], np.float32)  # image(image(a)) - b is parallel to mirror 2
    mv = np.array([[[5.0, 0, 0], [0, 0, 0.0]]], np.float32)
    mn = np.array([[[1.0, 0, 0], [0, 0, 1.0]]], np.float32)
    exp = orc.image_method(a, b, mv, mn)
    ta = torch.tensor(a, device="cuda", requires_grad=True)
    tb = torch.tensor(b, device="cuda", requires_grad=True)
    tmv = torch.tensor(mv, device="cuda", requires_grad=True)
    tmn = torch.tensor(mn, device="cuda", requires_grad=True)
    got = G.image_method(ta, tb, tmv, tmn)
    np.testing.assert_array_equal(_bits(_np(got)), _bits(exp))
    assert np.isinf(exp).all()  # parallel at mirror 2 -> inf, propagated to mirror 1
    torch.where(torch.isfinite(got), got, torch.zeros_like(got)).sum().backward()
    for t in (ta, tb, tmv, tmn):
        assert not torch.isnan(t.grad).any()


@pytest.mark.parametrize("k", [1, 2, 3, 5])
def test_image_method_vjp_vs_autograd(G, rng, k):
    """Hand-written VJP kernel vs torch.autograd over the torch restatement of the reference.

    Truth = float64 autograd.  Random mirror chains contain a few near-parallel (ill-conditioned)
    samples where ANY float32 evaluation is off by cond*eps, so the bar is per sample:
    error <= 1e-5 relative, or no worse than 4x the error of plain float32 autograd (the
    reference's own precision); and as many samples as float32 autograd must meet 1e-5 outright."""
    B = 257
    a = rng.normal(size=(B, 3)) * 3
    b = rng.normal(size=(B, 3)) * 3
    mv = rng.normal(size=(B, k, 3))
    mn = rng.normal(size=(B, k, 3))
    mn /= np.linalg.norm(mn, axis=-1, keepdims=True)
    w = rng.normal(size=(B, k, 3))
    a, b, mv, mn, w = (x.astype(np.float32) for x in (a, b, mv, mn, w))

    def grads(dtype, device, fn):
        ins = [torch.tensor(x, dtype=dtype, device=device, requires_grad=True) for x in (a, b, mv, mn)]
        (fn(*ins) * torch.tensor(w, dtype=dtype, device=device)).sum().backward()
        return [t.grad.detach().cpu().double().numpy().reshape(B, -1) for t in ins]

    g64 = grads(torch.float64, "cpu", torch_ref.image_method)
    g32 = grads(torch.float32, "cpu", torch_ref.image_method)
    ggpu = grads(torch.float32, "cuda", G.image_method)
    for got, ref32, truth, name in zip(ggpu, g32, g64, ("from", "to", "mirror_vertices", "mirror_normals")):
        scale = np.abs(truth).max(axis=1) + 1e-30
        err_gpu = np.abs(got - truth).max(axis=1) / scale
        err_ref = np.abs(ref32 - truth).max(axis=1) / scale
        assert np.isfinite(got).all(), name
        assert (err_gpu <= np.maximum(RTOL, 4 * err_ref)).all(), (name, err_gpu.max(), err_ref.max())
        # as many samples within 1e-5 as plain float32 autograd manages (long random chains are
        # ill-conditioned more often)
        assert (err_gpu <= RTOL).mean() >= (err_ref <= RTOL).mean() - 0.05, (
            name, (err_gpu <= RTOL).mean(), (err_ref <= RTOL).mean())


# ------------------------------------------------------------------ fused trace ----
def _scene(G, tb, tx, rx, assume_quads=False, mask=None):
    mesh = G.Mesh(tb["vertices"], tb["triangles"], mask=mask, assume_quads=assume_quads)
    return G.Scene(np.asarray(tx, np.float32), np.asarray(rx, np.float32), mesh)


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("mesh_mask", [False, True])
def test_two_buildings_goldens(G, goldens, two_buildings, order, assume_quads, mesh_mask):
    """differt/tests/geometry/test_scene.py:116-260 (exhaustive rows) through Scene.trace_paths,
    plus bit parity of every dense output with the oracle, plus dense == compact."""
    g = goldens["advanced_path_tracing_example"]
    exp = g["orders"][str(order)]
    if order == 4 and not assume_quads:
        pytest.skip("292k candidates x dense layout: covered by the compact test below")
    mask = np.ones(two_buildings["triangles"].shape[0], bool) if mesh_mask else None
    scene = _scene(G, two_buildings, g["tx"], g["rx"], assume_quads, mask)
    got = scene.trace_paths(order)
    C = got.mask.shape[-1]
    assert tuple(got.vertices.shape) == (C, order + 2, 3) and tuple(got.objects.shape) == (C, order + 2)
    exp_objects = np.asarray(exp["objects"], np.int32)
    if assume_quads:
        exp_objects = exp_objects - exp_objects % 2
    exp_vertices = orc.assemble_path(np.asarray(g["tx"], np.float32),
                                     np.asarray(exp["path_vertices"], np.float32).reshape(1, order, 3),
                                     np.asarray(g["rx"], np.float32))
    np.testing.assert_array_equal(_np(got.masked_objects), exp_objects)
    np.testing.assert_allclose(_np(got.masked_vertices), exp_vertices, rtol=g["rtol"])
    assert int(got.num_valid_paths) == 1
    # oracle, every candidate
    n_prim = scene.mesh.num_primitives
    cand = orc.generate_all_path_candidates(n_prim, order).astype(np.int32) * (2 if assume_quads else 1)
    o = orc.trace_path_candidates(two_buildings["vertices"], two_buildings["triangles"], g["tx"], g["rx"],
                                  cand, mask=mask, assume_quads=assume_quads)
    np.testing.assert_array_equal(_np(got.mask).reshape(-1), o["mask"].reshape(-1))
    np.testing.assert_array_equal(_np(got.objects).reshape(-1), o["objects"].reshape(-1))
    np.testing.assert_array_equal(_bits(_np(got.vertices)).reshape(-1), _bits(o["vertices"]).reshape(-1))
    # compact == masked dense
    # (round 6: for orders 1..3 the default compact tracer IS the pruned search, drt_trace_paths_beam; `literal=True` evaluates
    # every candidate with the filter kernel -- the same rows and the same rank keys either way)
    for kw in ({}, {"literal": True}):
        cp = scene.trace_paths(order, compact=True, **kw)
        np.testing.assert_array_equal(_np(cp.objects), _np(got.masked_objects))
        np.testing.assert_array_equal(_bits(_np(cp.vertices)), _bits(_np(got.masked_vertices)))
        np.testing.assert_array_equal(_np(cp.keys), np.flatnonzero(o["mask"].reshape(-1)))


def test_two_buildings_order4_compact(G, goldens, two_buildings):
    """test_scene.py:146-160 order 4 without assume_quads: 292 008 candidates, GPU-unranked."""
    g = goldens["advanced_path_tracing_example"]
    exp = g["orders"]["4"]
    scene = _scene(G, two_buildings, g["tx"], g["rx"])
    cp = scene.trace_paths(4, compact=True)
    np.testing.assert_array_equal(_np(cp.objects), np.asarray(exp["objects"], np.int32))
    exp_vertices = orc.assemble_path(np.asarray(g["tx"], np.float32),
                                     np.asarray(exp["path_vertices"], np.float32).reshape(1, 4, 3),
                                     np.asarray(g["rx"], np.float32))
    np.testing.assert_allclose(_np(cp.vertices), exp_vertices, rtol=g["rtol"])
    cand = orc.generate_all_path_candidates(24, 4).astype(np.int32)
    o = orc.trace_path_candidates(two_buildings["vertices"], two_buildings["triangles"], g["tx"], g["rx"], cand)
    np.testing.assert_array_equal(_np(cp.keys), np.flatnonzero(o["mask"].reshape(-1)))


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_canyon_multi_tx_rx(G, rng, order, assume_quads):
    """3 TX x 5 RX in a masked street canyon (tests/conftest.py): dense outputs bit-identical to the
    oracle, compact output = masked dense in masked_vertices order (geometry/_paths.py:274-297)."""
    from conftest import canyon_case

    V, Tr, mask, tx, rx, cand = canyon_case(rng, order, assume_quads)
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand, mask=mask, assume_quads=assume_quads)
    scene = G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads))
    got = scene.trace_paths(path_candidates=cand)
    assert tuple(got.mask.shape) == (3, 5, cand.shape[0])
    np.testing.assert_array_equal(_np(got.mask), o["mask"])
    np.testing.assert_array_equal(_np(got.objects), o["objects"])
    np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(o["vertices"]))
    assert o["mask"].sum() >= 40, "test scene produced too few valid paths: not a meaningful parity case"
    cp = scene.trace_paths(path_candidates=cand, compact=True)
    np.testing.assert_array_equal(_np(cp.keys), np.flatnonzero(o["mask"].reshape(-1)))
    np.testing.assert_array_equal(_np(cp.objects), o["objects"].reshape(-1, order + 2)[o["mask"].reshape(-1)])
    np.testing.assert_array_equal(_bits(_np(cp.vertices)),
                                  _bits(o["vertices"].reshape(-1, order + 2, 3)[o["mask"].reshape(-1)]))


def test_rank_window_and_disconnect(G, goldens, two_buildings):
    """GPU unranking over rank windows == table; disconnect_inactive_triangles == mask
    (test_scene.py:585-724)."""
    g = goldens["advanced_path_tracing_example"]
    rng = np.random.default_rng(5)
    mask = rng.random(24) > 0.3
    mask[[8, 9, 22]] = True
    scene = _scene(G, two_buildings, g["tx"], g["rx"], mask=mask)
    full = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2)
    a = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, 0, 200)
    b = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, 200, None)
    keys = np.concatenate([_np(a.keys), _np(b.keys) + 200])
    np.testing.assert_array_equal(keys, _np(full.keys))
    np.testing.assert_array_equal(np.concatenate([_np(a.objects), _np(b.objects)]), _np(full.objects))
    dis = G.ExhaustivePathTracer(disconnect_inactive_triangles=True)
    d = dis.trace_rank_range_literal(scene, 2)
    np.testing.assert_array_equal(_np(d.objects), _np(full.objects))
    np.testing.assert_array_equal(_bits(_np(d.vertices)), _bits(_np(full.vertices)))
    # candidate tables: GPU fill == host unranking == oracle odometer
    cands, types = dis.generate_path_candidates(scene, 2)
    act = np.flatnonzero(mask)
    np.testing.assert_array_equal(_np(cands), act[orc.generate_all_path_candidates(len(act), 2)])
    assert (_np(types) == 0).all()
    sub = scene.with_mesh(scene.mesh.masked())
    s = sub.trace_paths(2, compact=True)
    np.testing.assert_array_equal(_bits(_np(s.vertices)), _bits(_np(full.vertices)))


def test_chunks_padding_empty(G, goldens, two_buildings):
    """chunk iteration (SC:735-751), -1 padded rows (SV:912-918), no candidates (SV:566-573),
    empty mesh (test_scene.py:444-534)."""
    g = goldens["advanced_path_tracing_example"]
    scene = _scene(G, two_buildings, g["tx"], g["rx"])
    whole = scene.trace_paths(2)
    chunks = list(scene.trace_paths(2, chunk_size=100))
    assert len(chunks) == -(-552 // 100)
    np.testing.assert_array_equal(np.concatenate([_np(c.mask) for c in chunks], axis=-1), _np(whole.mask))
    tracer = G.ExhaustivePathTracer()
    padded = list(tracer.generate_path_candidates_chunks_iter(scene, 2, chunk_size=100, pad_chunks=True))
    assert all(c.shape == (100, 2) for c, _ in padded) and (_np(padded[-1][0])[-1] == -1).all()
    last = tracer.trace_path_candidates(scene, *padded[-1])
    assert not _np(last.mask)[..., 52:].any() and (_np(last.vertices)[..., 52:, :, :] == 0).all()
    none = scene.trace_paths(path_candidates=np.zeros((0, 3), np.int32))
    assert tuple(none.vertices.shape) == (0, 5, 3)
    empty = G.Scene(g["tx"], g["rx"], G.Mesh.empty())
    los = empty.trace_paths(0)
    assert _np(los.mask).tolist() == [True] and _np(los.objects).tolist() == [[0, 0]]
    assert tuple(empty.trace_paths(2).vertices.shape) == (0, 4, 3)
    with pytest.raises(ValueError):
        scene.trace_paths()


def test_tx_rx_grids_shape(G, two_buildings):
    """test_scene.py:536-562."""
    scene = _scene(G, two_buildings, [0, 0, 1], [1, 1, 1]).with_transmitters_grid(3, 2).with_receivers_grid(4, 5)
    assert tuple(scene.transmitters.shape) == (2, 3, 3) and tuple(scene.receivers.shape) == (5, 4, 3)
    p = scene.trace_paths(1)
    assert tuple(p.mask.shape) == (2, 3, 5, 4, 24)
    assert tuple(p.vertices.shape) == (2, 3, 5, 4, 24, 3, 3)


def test_config1_box(G):
    """BASELINE configs[0]: 1 TX, 1 RX, 12-triangle box, order 1."""
    V, Tr = orc.box_mesh(with_top=True)
    tx, rx = [0.1, -0.2, 0.05], [-0.3, 0.25, -0.1]
    cand = orc.generate_all_path_candidates(12, 1).astype(np.int32)
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand)
    scene = G.Scene(tx, rx, G.Mesh.box(with_top=True))
    np.testing.assert_array_equal(_np(scene.mesh.vertices), V)
    np.testing.assert_array_equal(_np(scene.mesh.triangles), Tr)
    np.testing.assert_array_equal(_bits(_np(scene.mesh.normals)), _bits(orc.mesh_normals(orc.triangle_vertices(V, Tr))))
    got = scene.trace_paths(1)
    np.testing.assert_array_equal(_np(got.mask), o["mask"].reshape(-1))
    np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(o["vertices"].reshape(-1, 3, 3)))
    q = scene.set_assume_quads().trace_paths(1)
    assert int(q.num_valid_paths) == 6


def test_config1_box_beam_raw_call_with_the_queried_workspace(G):
    """configs[0] at orders 0..3 through drt_trace_paths_beam called directly with drt_trace_beam_workspace_size's
    answer for the scene (default capacities: < 64 MB for the box) == the exhaustive tracer."""
    import ctypes as C

    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream
    from differt_amd.geometry._solvers import _params

    scene = G.Scene([0.1, -0.2, 0.05], [-0.3, 0.25, -0.1], G.Mesh.box(with_top=True))
    tx = scene.transmitters.reshape(-1, 3).contiguous()
    rx = scene.receivers.reshape(-1, 3).contiguous()
    L = _lib.load()
    for order in (0, 1, 2, 3):
        ex = G.ExhaustivePathTracer().trace_rank_range_literal(scene, order)
        mp = 256
        nbytes = L.drt_trace_beam_workspace_size(1, 1, 12, order, None, mp)
        assert nbytes < 64 << 20
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        keys = torch.empty(mp, dtype=torch.int64, device="cuda")
        verts = torch.empty((mp, order + 2, 3), dtype=torch.float32, device="cuda")
        objs = torch.empty((mp, order + 2), dtype=torch.int32, device="cuda")
        nv = C.c_int64(0)
        params = _params(None, None, None)
        _lib.call("drt_trace_paths_beam", scene.mesh.handle().h, C.byref(params), None, ptr(tx), 1, ptr(rx), 1, order, mp,
                  ptr(keys), ptr(verts), ptr(objs), C.byref(nv), ptr(ws), nbytes, stream())
        n = int(nv.value)
        assert n == ex.objects.shape[0] and (order == 0 or n > 0)
        np.testing.assert_array_equal(_np(objs[:n]), _np(ex.objects))
        np.testing.assert_array_equal(_bits(_np(verts[:n])), _bits(_np(ex.vertices)))


def test_beam_order0_output_overflow_regrows(G, rng):
    """line-of-sight pairs beyond max_paths -- the order-0 branch forwards to the compact tracer, whose
    overflow must come back as the regrowable "raise max_paths" condition, not as a hard CapacityError."""
    tx = rng.uniform(-1, 1, (40, 3)).astype(np.float32) + np.array([0, 0, 50], np.float32)
    rx = rng.uniform(-1, 1, (50, 3)).astype(np.float32) + np.array([0, 0, 60], np.float32)
    scene = G.Scene(tx, rx, G.Mesh.box(with_top=True))
    ex = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 0)
    assert ex.objects.shape[0] == 2000
    bp = G.ExhaustivePathTracer().trace_beam_pruned(scene, 0, max_paths=64)
    assert torch.equal(bp.objects, ex.objects) and torch.equal(bp.vertices, ex.vertices)
    cp = scene.trace_paths(0, solver="beam", max_paths=100)
    assert cp.objects.shape[0] == 2000


# ------------------------------------------------------------------ gradients ----
@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("compact", [False, True])
def test_trace_grad_vs_autograd(G, goldens, two_buildings, order, compact):
    """d(sum of valid path lengths)/d(tx, rx, mesh vertices): HIP VJP vs torch.autograd over the
    torch restatement, <= 1e-5 relative (BASELINE north-star)."""
    g = goldens["advanced_path_tracing_example"]
    rng = np.random.default_rng(11)
    tx = (np.asarray(g["tx"], np.float32) + rng.normal(size=(3, 3)).astype(np.float32) * 0.3)
    rx = (np.asarray(g["rx"], np.float32) + rng.normal(size=(2, 3)).astype(np.float32) * 0.3)
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    cand = orc.generate_all_path_candidates(24, order).astype(np.int32)
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand)
    valid = o["mask"].reshape(-1)
    assert valid.sum() > 0

    def length(v):
        return torch.sqrt((torch.diff(v, dim=-2) ** 2).sum(-1)).sum()

    vt = torch.tensor(V, dtype=torch.float64, requires_grad=True)
    txt = torch.tensor(tx, dtype=torch.float64, requires_grad=True)
    rxt = torch.tensor(rx, dtype=torch.float64, requires_grad=True)
    full = torch_ref.trace_vertices(vt, torch.tensor(Tr, dtype=torch.long), txt, rxt, torch.tensor(cand, dtype=torch.long))
    length(full.reshape(-1, order + 2, 3)[torch.tensor(valid)]).backward()

    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    rxg = torch.tensor(rx, device="cuda", requires_grad=True)
    vg = torch.tensor(V, device="cuda", requires_grad=True)
    scene = G.Scene(txg, rxg, G.Mesh(vg, Tr))
    paths = scene.trace_paths(order, compact=compact)
    got_v = paths.vertices if compact else paths.masked_vertices
    assert got_v.shape[0] == valid.sum()
    length(got_v).backward()
    for got, exp, name in ((txg, txt, "tx"), (rxg, rxt, "rx"), (vg, vt, "vertices")):
        e = exp.grad.numpy()
        scale = np.abs(e).max()
        np.testing.assert_allclose(_np(got.grad), e, rtol=RTOL, atol=RTOL * scale, err_msg=name)


def test_first_hit_box_and_jacobians(G, goldens):
    """differt/tests/geometry/test_mesh.py:1984-2073: mesh-bound queries == free functions on
    Mesh.box(2,2,2); Jacobians of t w.r.t. origins, directions, vertices at 1e-5."""
    g = goldens["first_hit_box"]
    mesh = G.Mesh.box(*g["box"])
    rng = np.random.default_rng(3)
    o = rng.uniform(-5, 5, (10, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (10, 3)).astype(np.float32)
    tv = _np(mesh.triangle_vertices)
    np.testing.assert_array_equal(_np(mesh.ray_intersect_any_triangle(o, d)), orc.ray_intersect_any_triangle(o, d, tv))
    ei, et = orc.first_triangle_hit_by_ray(o, d, tv)
    gi, gt = mesh.first_triangle_hit_by_ray(o, d)
    np.testing.assert_array_equal(_np(gi), ei)
    np.testing.assert_array_equal(_np(gt), et)

    o = np.asarray(g["ray_origins"], np.float32)
    d = np.asarray(g["ray_directions"], np.float32)
    V, Tr = _np(mesh.vertices), _np(mesh.triangles)
    faces, _ = orc.first_triangle_hit_by_ray(o, d, tv)
    assert (faces >= 0).all()
    vt = torch.tensor(V, dtype=torch.float64, requires_grad=True)
    ot = torch.tensor(o, dtype=torch.float64, requires_grad=True)
    dt = torch.tensor(d, dtype=torch.float64, requires_grad=True)
    jac_ref = torch.autograd.functional.jacobian(
        lambda oo, dd, vv: torch_ref.differentiable_distance(vv, torch.tensor(Tr, dtype=torch.long), oo, dd,
                                                             torch.tensor(faces, dtype=torch.long)),
        (ot, dt, vt))
    og = torch.tensor(o, device="cuda", requires_grad=True)
    dg = torch.tensor(d, device="cuda", requires_grad=True)
    vg = torch.tensor(V, device="cuda", requires_grad=True)

    def fun(oo, dd, vv):
        return G.Mesh(vv, Tr).first_triangle_hit_by_ray(oo, dd)[1]

    jac_got = torch.autograd.functional.jacobian(fun, (og, dg, vg))
    for jg, jr in zip(jac_got, jac_ref):
        np.testing.assert_allclose(_np(jg), jr.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("disconnect", [False, True])
def test_generate_path_candidates_gpu_fill(G, two_buildings, goldens, assume_quads, disconnect):
    """ExhaustivePathTracer.generate_path_candidates (GPU unranking) == the host enumerator == the
    oracle's odometer (test_scene.py:334-364 candidates round trip; _solvers.py:803-848)."""
    g = goldens["advanced_path_tracing_example"]
    rng = np.random.default_rng(4)
    mask = rng.random(24) > 0.3
    if assume_quads:
        mask[1::2] = mask[0::2]
    scene = _scene(G, two_buildings, g["tx"], g["rx"], assume_quads, mask)
    tracer = G.ExhaustivePathTracer(disconnect_inactive_triangles=disconnect)
    for order in (0, 1, 2, 3):
        cands, types = tracer.generate_path_candidates(scene, order)
        prim_mask = (mask[0::2] & mask[1::2]) if assume_quads else mask
        nodes = np.flatnonzero(prim_mask) if disconnect else np.arange(len(prim_mask))
        exp = nodes[orc.generate_all_path_candidates(len(nodes), order)] if order else np.zeros((1, 0), np.int64)
        exp = exp * (2 if assume_quads else 1)
        assert tuple(cands.shape) == exp.shape and cands.dtype == torch.int32
        np.testing.assert_array_equal(_np(cands), exp)
        assert (_np(types) == 0).all()
        # tracing the generated table == tracing the rank range
        a = tracer.trace_path_candidates_compact(scene, cands)
        b = tracer.trace_rank_range_literal(scene, order)
        assert torch.equal(a.objects, b.objects) and torch.equal(a.vertices, b.vertices)


def test_compute_paths_deprecated_front_end(G, goldens, two_buildings):
    """_scene.py:1046-1248: deprecated dispatcher (DeprecationWarning, same results, ValueError)."""
    g = goldens["advanced_path_tracing_example"]
    scene = _scene(G, two_buildings, g["tx"], g["rx"])
    ref = scene.trace_paths(2)
    with pytest.warns(DeprecationWarning):
        a = scene.compute_paths(2)
    with pytest.warns(DeprecationWarning):
        b = scene.compute_paths(2, method="hybrid", num_rays=100_000)
    with pytest.warns(DeprecationWarning):
        c = scene.compute_paths(1, method="sbr", num_rays=50_000, max_dist=1e-1)
    assert torch.equal(a.mask, ref.mask) and torch.equal(a.vertices, ref.vertices)
    assert torch.equal(b.masked_objects, ref.masked_objects)
    assert isinstance(c, G.LaunchedPaths) and bool(c.mask.any())
    with pytest.warns(DeprecationWarning), pytest.raises(ValueError):
        scene.compute_paths(method="sbr")
    with pytest.warns(DeprecationWarning), pytest.raises(ValueError):
        scene.compute_paths()


def test_normalize_and_assemble_path(G, rng):
    """_utils.py:29-72 (zero vectors -> 0, length 0) and :514-565, bit-exact vs the oracle."""
    v = rng.normal(size=(7, 5, 3)).astype(np.float32) * 10
    v[0, 0] = 0
    got, lens = G.normalize(v)
    exp, elens = orc.normalize(v)
    np.testing.assert_array_equal(_bits(_np(got)), _bits(exp))
    np.testing.assert_array_equal(_bits(_np(lens)), _bits(elens))
    assert (_np(got)[0, 0] == 0).all() and _np(lens)[0, 0] == 0
    assert tuple(G.normalize(v, keepdims=True)[1].shape) == (7, 5, 1)
    a, m, b = rng.normal(size=(4, 1, 3)), rng.normal(size=(1, 6, 2, 3)), rng.normal(size=(3,))
    np.testing.assert_array_equal(_np(G.assemble_path(a, m, b)), orc.assemble_path(a, m, b))
    assert tuple(G.assemble_path(a[:, 0], b).shape) == (4, 2, 3)


# ------------------------------------------------------------------ hybrid tracer, GPU-resident ----
@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("mesh_mask", [False, True])
def test_hybrid_rank_range_equals_host_enumeration(G, goldens, two_buildings, order, assume_quads, mesh_mask):
    """The pruned product space F x N^(order-2) x L unranked on the GPU (drt_candidates.first_map /
    last_map) yields exactly the valid paths, in the same order, as tracing the table that the host
    DiGraph enumerates (reference _solvers.py:996-1056); rank windows concatenate to the whole."""
    g = goldens["advanced_path_tracing_example"]
    rng = np.random.default_rng(9)
    mask = None
    if mesh_mask:
        mask = rng.random(two_buildings["triangles"].shape[0]) > 0.25
        if assume_quads:
            mask[1::2] = mask[0::2]
    tx = np.asarray(g["tx"], np.float32) + np.array([[0, 0, 0], [3, -2, 1]], np.float32)
    rx = np.asarray(g["rx"], np.float32) + np.array([[0, 0, 0], [-2, 1, 0.5], [4, 3, 1]], np.float32)
    scene = _scene(G, two_buildings, tx, rx, assume_quads, mask)
    solver = G.HybridPathTracer(num_rays=200_000)
    cands, _ = solver.generate_path_candidates(scene, order)
    ref = solver.trace_path_candidates_compact(scene, cands)
    got = scene.trace_paths(order, solver=solver, compact=True)
    np.testing.assert_array_equal(_np(got.objects), _np(ref.objects))
    np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(_np(ref.vertices)))
    total = solver.num_path_candidates(scene, order)
    assert total >= cands.shape[0]  # the product space still counts equal-neighbour tuples
    if order >= 2:
        n_first, n_last = (int(x.shape[0]) for x in solver._visible_sets(scene)[:2])
        assert 0 < n_first and 0 < n_last
    cuts = [0, total // 3, total // 3 + 1, total]
    parts = [solver.trace_rank_range(scene, order, lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
    n_valid = sum(p.objects.shape[0] for p in parts)
    assert n_valid == ref.objects.shape[0]
    if n_valid:
        # the same set of paths, whatever the window cut
        both = np.concatenate([_np(p.objects) for p in parts])
        assert sorted(map(tuple, both.tolist())) == sorted(map(tuple, _np(ref.objects).tolist()))
    # gradients flow through the product-space path like through any compact trace
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    sc = G.Scene(txg, rx, scene.mesh)
    p = sc.trace_paths(order, solver=solver, compact=True)
    p.vertices.square().sum().backward()
    assert torch.isfinite(txg.grad).all()


def test_hybrid_rank_range_empty_sets(G):
    """No primitive visible from the receiver (it sits inside a closed box, the transmitter outside)."""
    outer = G.Mesh.box(2.0, 2.0, 2.0, with_top=True)
    scene = G.Scene([5.0, 0.0, 0.0], [0.0, 0.0, 0.0], outer)
    solver = G.HybridPathTracer(num_rays=50_000)
    for order in (1, 2, 3):
        p = solver.trace_rank_range(scene, order)
        ref = solver.trace_path_candidates_compact(scene, solver.generate_path_candidates(scene, order)[0])
        assert p.objects.shape[0] == ref.objects.shape[0]


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("world", [2, 5])
def test_beam_pruned_prefix_shards_partition_the_result(G, order, world):
    """Multi-GPU split of the beam-pruned tracer (distributed.trace_beam_pruned_sharded), emulated on one GPU:
    the shards' valid paths are disjoint and their key-sorted union is the unsharded result bit for bit."""
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(24, pitch=30.0, seed=5)
    tx, rx = S.manhattan_tx_rx(c, h, 3, 6, seed=6, pitch=30.0)
    mesh = G.Mesh(V, Tr)
    scene = G.Scene(torch.as_tensor(tx, device="cuda"), torch.as_tensor(rx, device="cuda"), mesh)
    tracer = G.ExhaustivePathTracer()
    full = tracer.trace_beam_pruned(scene, order)
    parts = [tracer.trace_beam_pruned(scene, order, prefix_shard=(r, world)) for r in range(world)]
    keys = torch.cat([p.keys for p in parts])
    assert keys.shape[0] == full.keys.shape[0] and torch.unique(keys).shape[0] == keys.shape[0]
    perm = torch.argsort(keys)
    assert torch.equal(keys[perm], full.keys)
    assert torch.equal(torch.cat([p.objects for p in parts])[perm], full.objects)
    assert torch.equal(torch.cat([p.vertices for p in parts])[perm].view(torch.int32), full.vertices.view(torch.int32))
    if order == 2:
        assert full.keys.shape[0] > 0
    if order > 0:
        with pytest.raises(ValueError):
            tracer.trace_beam_pruned(scene, order, prefix_shard=(world, world))


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_beam_pruned_clustered_emit_equals_plain(G, order, assume_quads):
    """The clustered receiver stage (receivers in Morton clusters of 64 with bounding boxes, lane = receiver for the
    surviving (prefix, cluster) pairs) returns the rows of the plain one: identical paths, 150 receivers (a
    full cluster, a full one and a partial one) on a flat grid plus a few elevated ones."""
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(16, pitch=30.0, seed=11)
    tx, _ = S.manhattan_tx_rx(c, h, 2, 1, seed=12, pitch=30.0)
    ext = float(np.abs(V[:, :2]).max()) + 5.0
    gx, gy = np.meshgrid(np.linspace(-ext, ext, 12), np.linspace(-ext, ext, 12))
    rx = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 1.5)], axis=1).astype(np.float32)
    rx = np.concatenate([rx, np.asarray([[0.0, 0.0, 60.0], [ext, -ext, 35.0], [3.0, 4.0, 20.0], [-ext, 2.0, 80.0],
                                         [1.0, ext, 15.0], [7.0, -9.0, 45.0]], np.float32)])
    assert rx.shape[0] == 150
    mesh = G.Mesh(V, Tr, assume_quads=assume_quads)
    scene = G.Scene(torch.as_tensor(tx, device="cuda"), torch.as_tensor(rx, device="cuda"), mesh)
    tracer = G.ExhaustivePathTracer()
    a = tracer.trace_beam_pruned(scene, order, emit="plain")
    rows_plain = tracer.last_beam_stats["rows"]
    b = tracer.trace_beam_pruned(scene, order, emit="clustered")
    assert tracer.last_beam_stats["rows"] == rows_plain  # the same candidate rows, not just the same survivors
    assert torch.equal(a.keys, b.keys) and torch.equal(a.objects, b.objects)
    assert torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32))
    if order <= 2:
        assert a.keys.shape[0] > 0
    with pytest.raises(ValueError):
        tracer.trace_beam_pruned(scene, order, emit="nope")


def test_hybrid_visible_sets_memo_follows_the_end_points(G):
    """num_path_candidates() + trace_rank_range() share one visibility estimate, keyed on the end
    points BY VALUE: moving the transmitter in place must not return the stale sets."""
    box = G.Mesh.box(2.0, 2.0, 2.0, with_top=True)
    tx = torch.tensor([[5.0, 0.3, 0.1]], device="cuda")
    rx = torch.tensor([[-5.0, 0.2, 0.3]], device="cuda")
    solver = G.HybridPathTracer(num_rays=50_000)
    scene = G.Scene(tx, rx, box)
    a = solver._visible_sets(scene)
    assert solver._visible_sets(scene) is a  # memo hit
    first_a = a[0].cpu().tolist()
    tx.copy_(torch.tensor([[0.3, 5.0, 0.1]], device="cuda"))  # same storage, new position
    b = solver._visible_sets(scene)
    assert b is not a and b[0].cpu().tolist() != first_a


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_hybrid_trace_pairs_equals_exhaustive(G, goldens, two_buildings, order, assume_quads):
    """Per-pair visibility pruning (MI355X extension) finds the valid paths of the exhaustive tracer on
    the two-buildings scene, in the same (pair-major, lexicographic) order, with the same vertices."""
    g = goldens["advanced_path_tracing_example"]
    tx = np.asarray(g["tx"], np.float32) + np.array([[0, 0, 0], [3, -2, 1]], np.float32)
    rx = np.asarray(g["rx"], np.float32) + np.array([[0, 0, 0], [-2, 1, 0.5], [4, 3, 1]], np.float32)
    scene = _scene(G, two_buildings, tx, rx, assume_quads)
    solver = G.HybridPathTracer(num_rays=300_000)
    got = solver.trace_pairs(scene, order)  # one ragged launch for all pairs
    ref = scene.trace_paths(order, compact=True)
    np.testing.assert_array_equal(_np(got.objects), _np(ref.objects))
    np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(_np(ref.vertices)))
    for strategy in ("loop", "prefix", "ragged"):  # one launch per pair / lane = prefix / lane = row
        alt = G.HybridPathTracer(num_rays=300_000, pairs_strategy=strategy).trace_pairs(scene, order)
        np.testing.assert_array_equal(_np(alt.objects), _np(ref.objects), err_msg=strategy)
        np.testing.assert_array_equal(_bits(_np(alt.vertices)), _bits(_np(ref.vertices)), err_msg=strategy)
        if strategy != "loop" and order >= 2:
            np.testing.assert_array_equal(_np(alt.keys), _np(got.keys), err_msg=strategy)
    cached = solver.trace_pairs(scene, order, visibility=solver.estimate_visibility(scene))
    assert torch.equal(cached.objects, got.objects) and torch.equal(cached.vertices, got.vertices)
    if order >= 2:  # packed keys: (tx nrx + rx) n^order + sum_j m_j n^(order-1-j), ascending = masked_vertices order
        k = _np(got.keys)
        o = _np(got.objects).astype(np.int64)
        n = scene.mesh.num_primitives
        exp = o[:, 0] * rx.shape[0] + o[:, -1]
        for j in range(order):
            exp = exp * n + o[:, 1 + j] // (2 if assume_quads else 1)
        assert (np.diff(k) > 0).all() and np.array_equal(k, exp) and solver.last_num_evaluated > 0
    if order >= 2:
        assert solver.last_num_evaluated < 6 * G.ExhaustivePathTracer().num_path_candidates(scene, order)
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    p = G.HybridPathTracer(num_rays=300_000).trace_pairs(G.Scene(txg, rx, scene.mesh), order)
    q = G.Scene(torch.tensor(tx, device="cuda", requires_grad=True), rx, scene.mesh)
    e = q.trace_paths(order, compact=True)
    p.vertices.square().sum().backward()
    e.vertices.square().sum().backward()
    np.testing.assert_allclose(_np(txg.grad), _np(q.transmitters.grad), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ two more reference tests ----
@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_path_candidates_match_exhaustive(G, rng, order, assume_quads):
    """differt/tests/geometry/test_scene.py:334-364 (on the canyon scene): tracing the candidates an
    exhaustive trace returned gives the same valid paths; every valid candidate traced alone is valid."""
    from conftest import canyon_scene

    V, Tr = canyon_scene(rng)
    scene = G.Scene([-15.0, 1.0, 8.0], [12.0, -2.0, 3.0], G.Mesh(V, Tr)).set_assume_quads(assume_quads)
    expected = scene.trace_paths(order=order)
    cands = expected.objects[:, 1:-1]
    got = scene.trace_paths(path_candidates=cands)
    assert torch.equal(got.masked_vertices, expected.masked_vertices)
    valid = expected.masked().objects[:, 1:-1]
    assert valid.shape[0] > 0
    for c in valid:
        one = scene.trace_paths(path_candidates=c[None, :])
        assert one.mask.numel() == 1 and bool(one.mask.reshape(()))


@pytest.mark.parametrize("shapes", [((3,), (3,), (5, 3)), ((10, 3), (10, 3), (1, 3)), ((10, 3), (1, 3), (2, 3)),
                                    ((4, 1, 3), (1, 6, 3), (3, 3))])
def test_image_method_returns_vertices_on_mirrors(G, rng, shapes):
    """differt/tests/geometry/test_image_method.py:181-218: every image-method vertex lies in its mirror's
    plane (infinite vertices excluded)."""
    a, b = (rng.normal(size=s).astype(np.float32) for s in shapes[:2])
    mv = rng.normal(size=shapes[2]).astype(np.float32)
    mn = rng.normal(size=shapes[2]).astype(np.float32)
    mn /= np.linalg.norm(mn, axis=-1, keepdims=True)
    paths = _np(G.image_method(a, b, mv, mn))
    d = ((paths - mv) * mn).sum(-1)
    d = np.nan_to_num(d, posinf=0.0, neginf=0.0)
    np.testing.assert_allclose(d, 0.0, atol=1e-4)


# ------------------------------------------------------------------ Warp-semantics opt-in ----
def test_mesh_queries_warp_semantics_match_the_jax_operators(G):
    """Mirror of differt/tests/geometry/test_mesh.py:1984-2028: on Mesh.box(2, 2, 2) with 10 random
    rays the reference asserts its Warp-backed mesh queries EQUAL to the pure-JAX operators (any-hit
    exactly, first-hit indices exactly and t at rtol = atol = 1e-5).  Same assertion for the opt-in
    `semantics="warp"` ray preparation (origin offset hit_tol*|d| / max_t = |d|(1-2 hit_tol),
    _mesh.py:3065-3070; origin nudge 1e-5, _mesh.py:195-199) -- over many seeds instead of one."""
    mesh = G.Mesh.box(2.0, 2.0, 2.0)
    tv = mesh.triangle_vertices
    blocked_any = 0
    for seed in range(40):
        rng = np.random.default_rng(seed)
        o = rng.uniform(-5, 5, (10, 3)).astype(np.float32)
        d = rng.uniform(-1, 1, (10, 3)).astype(np.float32)
        expected = G.ray_intersect_any_triangle(o, d, tv)
        got = mesh.ray_intersect_any_triangle(o, d, semantics="warp")
        assert torch.equal(got, expected), seed
        blocked_any += int(expected.sum())
        ei, et = G.first_triangle_hit_by_ray(o, d, tv)
        gi, gt = mesh.first_triangle_hit_by_ray(o, d, semantics="warp")
        assert torch.equal(gi, ei), seed
        np.testing.assert_allclose(_np(gt), _np(et), rtol=1e-5, atol=1e-5)
        bi, bt = mesh.first_triangle_hit_by_ray(o, d, semantics="warp", accel="bvh")
        assert torch.equal(bi, ei) and torch.equal(bt.view(torch.int32), gt.view(torch.int32))
    assert blocked_any > 0
    # segment-style rays that START on a face (the case the offset exists for): leaving the face is not a hit
    o = torch.tensor([[1.0, 0.2, 0.1]], device="cuda")  # on the x = +1 face
    assert not bool(mesh.ray_intersect_any_triangle(o, torch.tensor([[2.0, 0.0, 0.0]], device="cuda"), semantics="warp"))
    assert bool(mesh.ray_intersect_any_triangle(o, torch.tensor([[-4.0, 0.0, 0.0]], device="cuda"), semantics="warp"))
    with pytest.raises(ValueError):
        mesh.ray_intersect_any_triangle(o, o, semantics="optix")


def test_mesh_first_hit_warp_semantics_jacobians(G):
    """test_mesh.py:2028-2072 for the opt-in mode: the nudge does not enter the gradient."""
    mesh = G.Mesh.box(2.0, 2.0, 2.0)
    o0 = torch.tensor([[0.0, 0.0, 3.0], [0.0, 3.0, 0.0], [3.0, 0.0, 0.0]], device="cuda")
    d0 = torch.tensor([[0.0, 0.0, -1.0], [0.0, -1.0, 0.0], [-1.0, 0.0, 0.0]], device="cuda")
    outs = []
    for sem in ("jax", "warp"):
        o, d = o0.clone().requires_grad_(True), d0.clone().requires_grad_(True)
        vl = mesh.vertices.detach().clone().requires_grad_(True)  # the leaf (Mesh keeps a reshaped view)
        m = mesh.with_vertices(vl)
        idx, t = m.first_triangle_hit_by_ray(o, d, semantics=sem)
        assert (idx >= 0).all()
        t.sum().backward()
        outs.append((t.detach(), o.grad, d.grad, vl.grad))
    np.testing.assert_allclose(_np(outs[1][0]), _np(outs[0][0]), rtol=1e-5, atol=1e-5)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        np.testing.assert_allclose(_np(b), _np(a), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ conservative beam pruning ----
def _assert_same_paths(a, b):
    assert a.objects.shape == b.objects.shape, (tuple(a.objects.shape), tuple(b.objects.shape))
    assert torch.equal(a.objects, b.objects)
    assert torch.equal(a.vertices.detach().view(torch.int32), b.vertices.detach().view(torch.int32))


@pytest.mark.parametrize("boxes,seed", [(30, 1), (60, 2), (100, 3), (150, 4)])
def test_beam_pruned_order3_equals_exhaustive(G, boxes, seed):
    """SURVEY.md section 7 hard part 1 (the reference's own answer is the sampled pruning of
    _solvers.py:1013-1056): on every city of scratch/order3_completeness.py (300..1500 triangles, 4 TX x
    16 RX) the geometrically pruned order-3 search returns EXACTLY the exhaustive tracer's valid paths
    -- same objects in the same order, vertex bits identical -- while evaluating a small fraction of the
    4 x 16 x n (n-1)^2 candidates."""
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(boxes, seed=seed)
    tx, rx = S.manhattan_tx_rx(c, h, 4, 16, seed=seed + 10)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer()
    ex = tracer.trace_rank_range_literal(scene, 3, max_survivors=1 << 24, max_paths=1 << 18)
    bp = tracer.trace_beam_pruned(scene, 3)
    _assert_same_paths(ex, bp)
    st_bvh = dict(tracer.last_beam_stats)
    for mapping in ("plain",):  # default "auto" = clustered; "plain" tests every (prefix, primitive) pair
        other = tracer.trace_beam_pruned(scene, 3, expansion=mapping)
        _assert_same_paths(ex, other)
        # box culling is the same test on a box with a bound of the candidates' own error: identical survivors at every
        # level but the last, where only the clustered mapping drops the children that cannot reach the receivers'
        # box -- children the receiver stage rejects anyway: identical ROWS
        pl = tracer.last_beam_stats
        assert st_bvh["levels"][:-1] == pl["levels"][:-1] and st_bvh["levels"][-1] <= pl["levels"][-1]
        assert st_bvh["rows"] == pl["rows"]
    n = Tr.shape[0]
    evaluated, total = tracer.last_beam_stats["rows"], 64 * n * (n - 1) ** 2
    assert ex.objects.shape[0] > 0 and evaluated < total / 20, (evaluated, total)
    # keys: (tx*nrx + rx) * n^3 + m1 n^2 + m2 n + m3, strictly increasing
    o = bp.objects.long()
    exp_keys = (o[:, 0] * 16 + o[:, 4]) * n ** 3 + o[:, 1] * n * n + o[:, 2] * n + o[:, 3]
    assert torch.equal(bp.keys, exp_keys) and bool((bp.keys[1:] > bp.keys[:-1]).all())


def test_beam_slice_hint_carried_between_calls(G, monkeypatch):
    """drt_beam_stats.next_probe_prefixes: the tracer object starts its next call on the same (mesh, order, end-point
    counts) with the slice size the previous call measured instead of a 4096-prefix probe -- fewer slices, the same paths
    (a hint only: a slice that overflows is retried smaller); DRT_BEAM_NO_HINT switches it off."""
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(700, seed=9)
    tx, rx = S.manhattan_tx_rx(c, h, 2, 16, seed=10)
    mesh = G.Mesh(V, Tr)
    tracer = G.ExhaustivePathTracer(accel="bvh")
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    first = tracer.trace_beam_pruned(scene, 2)
    s1 = dict(tracer.last_beam_stats)
    assert s1["chunks"] >= 2 and tracer._beam_probe_hints  # 2 x 3500 prefixes: a probe slice and the rest
    moved = G.Scene(torch.tensor(tx + np.float32(0.25), device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    second = tracer.trace_beam_pruned(moved, 2)
    s2 = dict(tracer.last_beam_stats)
    assert s2["chunks"] < s1["chunks"]
    monkeypatch.setenv("DRT_BEAM_NO_HINT", "1")
    ref = tracer.trace_beam_pruned(moved, 2)
    assert tracer.last_beam_stats["chunks"] == s1["chunks"]
    _assert_same_paths(ref, second)
    assert torch.equal(ref.keys, second.keys) and first.objects.shape[1] == 4
    # an absurd hint is harmless: the oversized slice is retried smaller
    huge = tracer.trace_beam_pruned(moved, 2, probe_prefixes=1 << 40, max_records=1 << 16)
    _assert_same_paths(ref, huge)


@pytest.mark.parametrize("accel", [None, "bvh"])
def test_dense_num_valid_paths_from_device_counters(G, accel):
    """The dense tracer leaves survivors / cleared-by-occlusion counters in its workspace (include/differt_amd.h);
    TracedPaths.num_valid_paths (reference _paths.py:264-272, read by the reference's harness after every call) uses
    them instead of reducing the mask -- the same number, through reshapes, and NOT after the mask was edited, sliced
    or replaced."""
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(30, seed=5)
    tx, rx = S.manhattan_tx_rx(c, h, 3, 7, seed=6)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
    total = 0
    for order in (0, 1, 2):
        for paths in ([scene.trace_paths(order=order, solver="exhaustive", accel=accel)] if order < 2 else
                      list(scene.trace_paths(order=order, solver="exhaustive", accel=accel, chunk_size=4096))[:6]):
            assert "_valid_count" in paths.__dict__ or paths.mask.numel() == 0
            expect = int(torch.count_nonzero(paths.mask))
            assert int(paths.num_valid_paths) == expect
            assert int(paths.reshape(-1).num_valid_paths) == expect
            if paths.mask.numel() > 1:
                sl = G.TracedPaths(paths.vertices.reshape(-1, order + 2, 3)[:1], paths.objects.reshape(-1, order + 2)[:1],
                                   paths.mask.reshape(-1)[:1])
                sl.__dict__["_valid_count"] = paths.__dict__.get("_valid_count")  # a slice must not trust the counters
                assert int(sl.num_valid_paths) == int(torch.count_nonzero(paths.mask.reshape(-1)[:1]))
                flat = paths.mask.reshape(-1)
                flat[0] = not bool(flat[0])  # in-place edit: version bump, back to the reduction
                assert int(paths.num_valid_paths) == int(torch.count_nonzero(paths.mask))
            total += expect
    assert total > 0


@pytest.mark.parametrize("order", [1, 2, 3])
def test_pair_blocks_through_the_c_abi_with_a_hand_built_table(G, rng, order):
    """DRT_CAND_PAIR_BLOCKS as a documented feature of drt_trace_paths_compact (include/differt_amd.h), independent of the
    beam search: a per-pair table whose blocks of 2^order rows enumerate the triangle choices 2 q_j + bit_j(c) of pair
    sequences -- sequences of valid paths and random ones, pairs that follow themselves (padding rows where a triangle
    would repeat), whole padding blocks, an empty (tx, rx) pair -- gives the same keys, objects and vertex bits with and
    without the bit."""
    import ctypes as C

    import differt_amd._lib as lib
    import synthetic_scenes as S
    from differt_amd._tensors import ptr, stream
    from differt_amd.geometry._solvers import _params

    V, Tr, c, h = S.manhattan(18, seed=31)
    ext = float(np.abs(V[:, :2]).max()) + 10
    gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
    Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
    V = np.concatenate((V, gv))
    tx, rx = S.manhattan_tx_rx(c, h, 2, 6, seed=32)
    tx[:, 2] = np.float32(float(h.max()) + 20.0)  # above every roof: ground and roof reflections exist at every order
    mesh = G.Mesh(V, Tr)
    txd, rxd = torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda")
    scene = G.Scene(txd, rxd, mesh)
    tracer = G.ExhaustivePathTracer()
    valid = tracer.trace_beam_pruned(scene, order)  # (builds the pair clusters: establishes the precondition)
    assert tracer.last_beam_stats["pair_mode"]
    nq, ntx, nrx, K = Tr.shape[0] // 2, len(tx), len(rx), order
    blocks = {p: [] for p in range(ntx * nrx)}
    for o in valid.objects.cpu().tolist():  # pair sequences of the valid paths
        blocks[o[0] * nrx + o[-1]].append(tuple(t // 2 for t in o[1:-1]))
    for p in range(ntx * nrx):
        if p == 3:
            blocks[p] = []  # an empty pair
            continue
        for _ in range(40):
            q = tuple(int(x) for x in rng.integers(0, nq, K))
            blocks[p].append(q)
        if K >= 2:
            blocks[p].append((nq - 1,) * K)  # the ground pair following itself
        blocks[p].append(None)  # a whole padding block
    rows, offsets = [], [0]
    for p in range(ntx * nrx):
        for q in blocks[p]:
            for cc in range(1 << K):
                if q is None:
                    rows.append([-1] * K)
                    continue
                ids = [2 * q[j] + ((cc >> (K - 1 - j)) & 1) for j in range(K)]
                rows.append([-1] * K if any(ids[j] == ids[j - 1] for j in range(1, K)) else ids)
        offsets.append(len(rows))
    table = torch.tensor(rows, dtype=torch.int32, device="cuda")
    pair_off = torch.tensor(offsets, dtype=torch.int64, device="cuda")
    L = lib.load()
    cap_s, cap_p = 1 << 16, 1 << 12
    nb = L.drt_trace_compact_workspace_size(cap_s, cap_p)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    params = _params(None, None, None)
    res = []
    for flag in (0, lib.DRT_CAND_PAIR_BLOCKS):
        cands = lib.Candidates()
        cands.table, cands.num_candidates, cands.order = table.data_ptr(), table.shape[0], K
        cands.pair_offsets = pair_off.data_ptr()
        cands.reserved = flag
        k = torch.empty(cap_p, dtype=torch.int64, device="cuda")
        v = torch.empty((cap_p, K + 2, 3), dtype=torch.float32, device="cuda")
        o = torch.empty((cap_p, K + 2), dtype=torch.int32, device="cuda")
        nv = C.c_int64(0)
        lib.call("drt_trace_paths_compact", mesh.handle().h, C.byref(params), ptr(txd), ntx, ptr(rxd), nrx, C.byref(cands),
                 cap_s, cap_p, ptr(k), ptr(v), ptr(o), C.byref(nv), ptr(ws), nb, stream())
        n = int(nv.value)
        res.append((n, k[:n].clone(), v[:n].clone(), o[:n].clone()))
    (n0, k0, v0, o0), (n1, k1, v1, o1) = res
    expected = {tuple(o) for o in valid.objects.cpu().tolist() if o[0] * nrx + o[-1] != 3}  # (pair 3 was emptied above)
    assert n0 == n1 and n0 >= len(expected) > 0  # (a valid sequence may appear twice: once found, once drawn)
    assert torch.equal(k0, k1) and torch.equal(o0, o1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32))
    assert expected <= set(map(tuple, o0.cpu().tolist()))


@pytest.mark.parametrize("order", [1, 2, 3])
def test_beam_pair_blocks_equal_row_by_row_trace(G, rng, order):
    """DRT_CAND_PAIR_BLOCKS (include/differt_amd.h): in coplanar-pair mode the filter stage evaluates the image chain once
    per surviving pair row and Moller-Trumbore against both triangles of every pair, instead of tracing the 2^order
    triangle rows one by one (rows="plain").  Same keys, objects and vertex bits, equal to the exhaustive tracer -- on
    box cities whose pair normals differ in the SIGN OF ZERO components (most axis-aligned walls), with a ground quad
    (a pair that may follow itself through its two triangles: ground - wall - ground), masks, and receivers placed on the
    diagonal of a wall's reflection so that both triangles of a pair can pass."""
    import synthetic_scenes as S

    total = 0
    for trial in range(4):
        boxes = int(rng.integers(5, 30))
        V, Tr, c, h = S.manhattan(boxes, pitch=float(rng.uniform(20, 45)), seed=int(rng.integers(1 << 30)))
        ext = float(np.abs(V[:, :2]).max()) + 10
        gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
        Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
        V = np.concatenate((V, gv))
        tx, rx = S.manhattan_tx_rx(c, h, 2, 24, seed=int(rng.integers(1 << 30)))
        tx[:, 2] = rng.uniform(2, 60, len(tx))
        # mirror image of a transmitter in the ground, seen from receivers on the ground quad's diagonal x = y
        rx[:6, 0] = rx[:6, 1]
        mask = None
        if trial % 2:
            mask = rng.random(Tr.shape[0]) > 0.15
            mask[1::2] = mask[0::2]
        mesh = G.Mesh(V, Tr, mask=mask)
        scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
        tracer = G.ExhaustivePathTracer()
        ex = tracer.trace_rank_range_literal(scene, order, max_survivors=1 << 22, max_paths=1 << 18)
        blocks = tracer.trace_beam_pruned(scene, order)
        st = dict(tracer.last_beam_stats)
        assert st["pair_mode"]
        plain = tracer.trace_beam_pruned(scene, order, rows="plain")
        assert tracer.last_beam_stats["rows"] == st["rows"]
        for got in (blocks, plain):
            _assert_same_paths(ex, got)
            assert torch.equal(got.keys, blocks.keys)
        total += ex.objects.shape[0]
    # the pair normals of these meshes do differ in zero signs (the case the kernel's argument is about)
    nrm = mesh.normals.view(torch.int32).reshape(-1, 2, 3)
    assert bool((nrm[:, 0] != nrm[:, 1]).any()) and bool((mesh.normals.reshape(-1, 2, 3)[:, 0] == mesh.normals.reshape(-1, 2, 3)[:, 1]).all())
    assert total > 0


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("masked", [False, True])
def test_beam_pruned_equals_exhaustive_quads_masks_orders(G, rng, order, assume_quads, masked):
    """Random small cities with a ground quad, primitives = triangles or quads, optional mask, orders 1..3,
    transmitters at random heights (also below roof level): pruned == exhaustive, and differentiable."""
    import synthetic_scenes as S

    for trial in range(3):
        boxes = int(rng.integers(4, 28))
        V, Tr, c, h = S.manhattan(boxes, pitch=float(rng.uniform(20, 45)), seed=int(rng.integers(1 << 30)))
        ext = float(np.abs(V[:, :2]).max()) + 10
        gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
        Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
        V = np.concatenate((V, gv))
        tx, rx = S.manhattan_tx_rx(c, h, 2, 5, seed=int(rng.integers(1 << 30)))
        tx[:, 2] = rng.uniform(2, 60, len(tx))
        mask = None
        if masked:
            mask = rng.random(Tr.shape[0]) > 0.15
            if assume_quads:
                mask[1::2] = mask[0::2]
        mesh = G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads)
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
        tracer = G.ExhaustivePathTracer()
        ex = tracer.trace_rank_range_literal(scene, order, max_survivors=1 << 22, max_paths=1 << 18)
        bp = tracer.trace_beam_pruned(scene, order)
        _assert_same_paths(ex, bp)
        rows_bvh = tracer.last_beam_stats["rows"]
        for mapping in ("plain",):
            _assert_same_paths(ex, tracer.trace_beam_pruned(scene, order, expansion=mapping))
            assert tracer.last_beam_stats["rows"] == rows_bvh  # same candidate rows
        if bp.objects.shape[0]:
            gref, = torch.autograd.grad(torch.sqrt((torch.diff(ex.vertices, dim=-2) ** 2).sum(-1)).sum(), txg)
            ggot, = torch.autograd.grad(torch.sqrt((torch.diff(bp.vertices, dim=-2) ** 2).sum(-1)).sum(), txg)
            np.testing.assert_allclose(_np(ggot), _np(gref), rtol=1e-5, atol=1e-6)


def test_beam_pruned_margin_only_widens_the_search(G):
    """A larger error unit (kappa) can only ADD rows, never lose a path."""
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(40, seed=9)
    tx, rx = S.manhattan_tx_rx(c, h, 2, 8, seed=19)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer()
    base = tracer.trace_beam_pruned(scene, 2)
    rows0 = tracer.last_beam_stats["rows"]
    wide = tracer.trace_beam_pruned(scene, 2, kappa=4096.0)
    rows1 = tracer.last_beam_stats["rows"]
    _assert_same_paths(base, wide)
    assert rows1 >= rows0 and tracer.last_beam_stats["unit_m"] > 0
    # small slices / tiny list capacities change nothing but the number of slices
    tiny = tracer.trace_beam_pruned(scene, 2, probe_prefixes=7, max_records=3000, max_rows=2000)
    _assert_same_paths(base, tiny)
    assert tracer.last_beam_stats["rows"] == rows0 and tracer.last_beam_stats["chunks"] > 3
