"""``Scene.trace_paths`` / ``launch_paths`` / ``compute_paths`` dispatch contract on the GPU path.

Mirrors differt/tests/geometry/test_scene.py: empty scenes :444-534, no candidate :650-679,
disconnect_inactive_triangles :681-725, hybrid == exhaustive-with-disconnect :727-757, kwargs
:919-935, solver interface coverage :943-1035, delegation/errors/warnings :1037-1099.
"""

from __future__ import annotations

import warnings
from collections.abc import Iterator
from contextlib import nullcontext as does_not_raise

import numpy as np
import pytest
import torch


pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


@pytest.fixture
def canyon(G, rng):
    from conftest import canyon_scene

    V, Tr = canyon_scene(rng)
    return G.Scene([-15.0, 1.0, 8.0], [12.0, -2.0, 3.0], G.Mesh(V, Tr))


def _np(x):
    return x.detach().cpu().numpy()


def _inner_outer(G, assume_quads):
    outer, inner = G.Mesh.box(6.0, 6.0, 6.0), G.Mesh.box(1.0, 1.0, 1.0)
    mesh = outer + inner
    mask = np.concatenate((np.ones(outer.num_triangles, bool), np.zeros(inner.num_triangles, bool)))
    mesh = mesh.set_mask(torch.as_tensor(mask, device=mesh.triangles.device))
    return G.Scene([-1.5, 0.0, 0.0], [1.5, 0.0, 0.0], mesh).set_assume_quads(assume_quads)


EMPTY_CASES = [
    (0, None, False, "exhaustive", does_not_raise),
    (0, 1000, False, "exhaustive", does_not_raise),
    (None, None, True, "exhaustive", does_not_raise),
    (0, None, True, "exhaustive", lambda: pytest.raises(ValueError, match="You must specify one of")),
    (None, 1000, True, "exhaustive", lambda: pytest.warns(UserWarning, match="Argument 'chunk_size' is ignored")),
    (None, None, True, "sbr", lambda: pytest.raises(ValueError, match="Argument 'order' is required")),
    (None, None, True, "hybrid", lambda: pytest.raises(ValueError, match="Argument 'order' is required")),
]


@pytest.mark.parametrize(("order", "chunk_size", "with_candidates", "method", "expectation"), EMPTY_CASES)
@pytest.mark.parametrize("assume_quads", [False, True])
def test_paths_on_empty_scene(G, rng, order, chunk_size, with_candidates, method, expectation, assume_quads):
    """test_scene.py:444-534: on an empty mesh the only path is the line of sight."""
    tx = rng.random((1, 3)).astype(np.float32)
    rx = rng.random((1, 3)).astype(np.float32)
    scene = G.Scene(tx, rx, G.Mesh.empty()).set_assume_quads(assume_quads)
    expected = np.stack((tx[0], rx[0]))[None, None, None]  # [1,1,1,2,3]
    cands = np.empty((1, 0), np.int32) if with_candidates else None
    with expectation():
        if method == "sbr":
            paths = scene.launch_paths(order=order, solver=G.SBRPathLauncher())
            assert isinstance(paths, G.LaunchedPaths)
        else:
            cls = G.ExhaustivePathTracer if method == "exhaustive" else G.HybridPathTracer
            got = scene.trace_paths(order=order, solver=cls(chunk_size=chunk_size), path_candidates=cands)
            paths = next(got) if isinstance(got, Iterator) else got
            assert isinstance(paths, G.TracedPaths)
        np.testing.assert_array_equal(_np(paths.vertices), expected)
        assert bool(paths.mask.all())


@pytest.mark.parametrize("assume_quads", [False, True])
def test_no_active_triangle_no_candidate(G, assume_quads):
    """test_scene.py:650-679."""
    mesh = G.Mesh.box(6.0, 6.0, 6.0)
    mesh = mesh.set_mask(torch.zeros(mesh.num_triangles, dtype=torch.bool, device=mesh.triangles.device))
    scene = G.Scene([-1.5, 0.0, 0.0], [1.5, 0.0, 0.0], mesh).set_assume_quads(assume_quads)
    paths = scene.trace_paths(order=1, solver=G.ExhaustivePathTracer(disconnect_inactive_triangles=True))
    assert paths.vertices.shape[0] == 0
    assert scene.trace_paths(order=1, solver="hybrid", num_rays=10_000).vertices.shape[0] == 0


@pytest.mark.parametrize("assume_quads", [False, True])
def test_disconnect_inactive_triangles(G, assume_quads):
    """test_scene.py:681-725: fewer candidates, the same valid paths."""
    scene = _inner_outer(G, assume_quads)
    full = scene.trace_paths(order=1, solver=G.ExhaustivePathTracer(disconnect_inactive_triangles=False))
    disc = scene.trace_paths(order=1, solver=G.ExhaustivePathTracer(disconnect_inactive_triangles=True))
    assert full.vertices.shape[0] > disc.vertices.shape[0]
    assert torch.equal(full.masked_vertices, disc.masked_vertices)
    assert torch.equal(full.masked_objects, disc.masked_objects)
    assert int(disc.num_valid_paths) > 0


@pytest.mark.parametrize("assume_quads", [False, True])
def test_hybrid_always_disconnects(G, assume_quads):
    """test_scene.py:727-757: hybrid == exhaustive with the inactive triangles disconnected (every
    face of the outer box is visible from both end points)."""
    scene = _inner_outer(G, assume_quads)
    hyb = scene.trace_paths(order=1, solver="hybrid")
    exh = scene.trace_paths(order=1, solver=G.ExhaustivePathTracer(disconnect_inactive_triangles=True))
    for name in ("vertices", "objects", "mask", "interaction_types"):
        assert torch.equal(getattr(hyb, name), getattr(exh, name)), name


def test_kwargs(G, canyon):
    """test_scene.py:919-935."""
    assert isinstance(canyon.trace_paths(order=1, chunk_size=10), Iterator)
    assert isinstance(canyon.trace_paths(order=1, solver="hybrid", chunk_size=10, num_rays=20_000), Iterator)
    assert isinstance(canyon.launch_paths(order=1, num_rays=500), G.LaunchedPaths)
    with pytest.raises(ValueError, match="solver_kwargs cannot be used"):
        canyon.trace_paths(order=1, solver=G.ExhaustivePathTracer(), chunk_size=10)
    with pytest.raises(ValueError, match="solver_kwargs cannot be used"):
        canyon.launch_paths(order=1, solver=G.SBRPathLauncher(), num_rays=10)


def test_chunked_iterators_cover_the_whole_table(G, canyon):
    whole = canyon.trace_paths(order=1)
    for solver in ("exhaustive", "hybrid"):
        kw = {"num_rays": 200_000} if solver == "hybrid" else {}
        chunks = list(canyon.trace_paths(order=1, solver=solver, chunk_size=7, **kw))
        mo = torch.cat([c.masked_objects for c in chunks])
        assert torch.equal(mo, whole.masked_objects), solver


def test_delegation_and_errors(G, canyon):
    """test_scene.py:1037-1067."""
    cands = np.zeros((1, 1), np.int32)
    for method in ("sbr", "hybrid"):
        with pytest.deprecated_call(), pytest.raises(ValueError, match="order' is required"):
            canyon.compute_paths(order=None, path_candidates=cands, method=method)
    with pytest.deprecated_call():
        assert isinstance(canyon.compute_paths(order=1, method="sbr", num_rays=500), G.LaunchedPaths)
    with pytest.deprecated_call():
        assert isinstance(canyon.compute_paths(order=1, method="hybrid", num_rays=500), G.TracedPaths)


def test_errors_and_warnings(G, canyon):
    """test_scene.py:1069-1099."""
    with pytest.raises(ValueError, match="Unknown solver"):
        canyon.trace_paths(order=1, solver="invalid")
    with pytest.raises(ValueError, match="Unknown solver"):
        canyon.launch_paths(order=1, solver="invalid")
    with pytest.warns(UserWarning, match="smoothing' is currently ignored"):
        canyon.trace_paths(order=1, solver=G.HybridPathTracer(smoothing_factor=0.1, num_rays=1000))
    with pytest.warns(UserWarning, match="chunk_size' is ignored"):
        canyon.trace_paths(path_candidates=np.zeros((1, 1), np.int32), solver=G.ExhaustivePathTracer(chunk_size=10))
    with pytest.raises(ValueError, match="order' is required"):
        canyon.launch_paths(order=None)
    with pytest.warns(DeprecationWarning), pytest.raises(ValueError, match="You must specify one of"):
        canyon.compute_paths(order=None, path_candidates=None)
    with warnings.catch_warnings():  # the plain call is silent
        warnings.simplefilter("error")
        canyon.trace_paths(order=1)


def test_solver_interfaces(G, canyon):
    """test_scene.py:943-1035: user-defined tracers / launchers plug into the same entry points."""

    class DummyTracer(G.AbstractPathTracer):
        def generate_path_candidates(self, scene, order, specular_reflection=True, diffuse_scattering=False):
            return torch.ones((5, 1), dtype=torch.int32), torch.zeros((5, 1), dtype=torch.int32)

        def trace_path_candidates(self, scene, path_candidates, interaction_types):
            n = path_candidates.shape[0]
            return G.TracedPaths(torch.zeros((1, 1, n, 3, 3)), torch.zeros((1, 1, n, 3), dtype=torch.int32),
                                 torch.zeros((1, 1, n), dtype=torch.bool),
                                 torch.zeros((1, 1, n, 1), dtype=torch.int32))

    tracer = DummyTracer()
    padded = list(tracer.generate_path_candidates_chunks_iter(canyon, order=1, chunk_size=2, pad_chunks=True))
    assert len(padded) == 3 and all(c.shape == (2, 1) for c, _ in padded) and int(padded[-1][0][-1, 0]) == -1
    unpadded = list(tracer.generate_path_candidates_chunks_iter(canyon, order=1, chunk_size=2, pad_chunks=False))
    assert len(unpadded) == 3 and unpadded[-1][0].shape == (1, 1)
    assert len(list(tracer.trace_paths(canyon, order=1, chunk_size=2, pad_chunks=True))) == 3
    assert isinstance(tracer.trace_paths(canyon, order=1), G.TracedPaths)
    assert tuple(canyon.trace_paths(order=1, solver=tracer).mask.shape) == (5,)

    class DummyLauncher(G.AbstractPathLauncher):
        max_dist = 1.0

        def launch_rays(self, scene):
            ntx = scene.transmitters.reshape(-1, 3).shape[0]
            z = torch.zeros((ntx, 10, 3), device=scene.transmitters.device)
            return z, z.clone()

    got = DummyLauncher().launch_paths(canyon, order=1)
    assert isinstance(got, G.LaunchedPaths) and not bool(got.masks.any())

    with pytest.raises(NotImplementedError):
        G.ExhaustivePathTracer().generate_path_candidates(canyon, order=[1, 2])
    with pytest.raises(NotImplementedError):
        G.HybridPathTracer(chunk_size=10).generate_path_candidates(canyon, order=[1, 2])
    with pytest.raises(NotImplementedError):
        G.HybridPathTracer(chunk_size=10).generate_path_candidates_chunks_iter(canyon, order=[1, 2])
    assert len(list(G.ExhaustivePathTracer().generate_path_candidates_chunks_iter(canyon, order=1, chunk_size=None))) == 1
    assert len(list(G.HybridPathTracer(chunk_size=None, num_rays=1000)
                    .generate_path_candidates_chunks_iter(canyon, order=1, chunk_size=None))) == 1


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("method", ["exhaustive", "sbr", "hybrid"])
def test_mesh_mask_matches_sub_mesh_without_mask(G, order, method):
    """test_scene.py:585-647: a masked inner box neither reflects nor occludes -- the valid paths equal
    those of the outer box alone (objects and vertices, every solver)."""
    outer, inner = G.Mesh.box(10.0, 10.0, 10.0), G.Mesh.box(4.0, 4.0, 4.0)
    mesh = outer + inner
    mask = np.concatenate((np.ones(outer.num_triangles, bool), np.zeros(inner.num_triangles, bool)))
    mesh = mesh.set_mask(torch.as_tensor(mask, device=mesh.triangles.device))
    tx, rx = [[-5.0, 0.0, 0.0]], [[5.0, 0.0, 0.0]]
    a, b = G.Scene(tx, rx, mesh), G.Scene(tx, rx, outer)
    if method == "sbr":
        got, exp = (s.launch_paths(order, solver="sbr", num_rays=200_000).masked() for s in (a, b))
    else:
        kw = {"num_rays": 200_000} if method == "hybrid" else {}
        got, exp = (s.trace_paths(order, solver=method, **kw).masked() for s in (a, b))
    assert torch.equal(got.objects, exp.objects)
    assert torch.equal(got.vertices, exp.vertices)
    if method != "sbr" and order <= 2:
        assert got.objects.shape[0] > 0


def test_trace_paths_beam_solver_equals_compact_exhaustive(G, two_buildings, goldens):
    """``Scene.trace_paths(order, solver="beam")``: the exhaustive solver's valid paths (reference scene goldens,
    differt/tests/geometry/test_scene.py:116-160) through the pruned search, same objects / vertex bits as ``compact=True``."""
    ex = goldens["advanced_path_tracing_example"]
    scene = G.Scene(np.asarray([ex["tx"]], np.float32), np.asarray([ex["rx"]], np.float32),
                    G.Mesh(two_buildings["vertices"], two_buildings["triangles"]))
    for order in (0, 1, 2, 3):
        a = scene.trace_paths(order, compact=True, literal=True)  # every candidate through the filter kernel
        b = scene.trace_paths(order, solver="beam", kappa=64.0)
        assert b.objects.cpu().numpy().tolist() == ex["orders"][str(order)]["objects"]
        assert torch.equal(a.objects, b.objects) and torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32))
    with pytest.raises(ValueError):
        scene.trace_paths(2, solver="beam", chunk_size=10)
    with pytest.raises(ValueError):
        scene.trace_paths(4, solver="beam")


@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("with_mask", [False, True])
@pytest.mark.parametrize("disconnect", [False, True])
def test_default_compact_tracer_is_the_pruned_search_with_rank_keys(G, canyon, assume_quads, with_mask, disconnect):
    """Round 6: ``Scene.trace_paths(order <= 3, compact=True)`` and a whole-space
    ``ExhaustivePathTracer.trace_rank_range`` run through ``drt_trace_paths_beam``; ``literal=True`` keeps the filter kernel
    (every candidate, as the reference enumerates them: _scene.py:650-764, _solvers.py:803-848).  Same objects, vertex bits,
    gradients AND keys -- ``(tx*num_rx + rx) * total + rank`` over the nodes of the candidate graph, inactive primitives
    removed from it with ``disconnect_inactive_triangles`` (_solvers.py:820-827)."""
    mesh = canyon.mesh.set_assume_quads(assume_quads)
    if with_mask:
        m = torch.ones(mesh.num_triangles, dtype=torch.bool, device="cuda")
        m[4:10] = False
        mesh = mesh.set_mask(m)
    tx = torch.tensor([[-15.0, 1.0, 8.0], [-12.0, -3.0, 5.0]], device="cuda")
    rx = torch.tensor([[12.0, -2.0, 3.0], [10.0, 2.0, 4.0], [0.0, 0.5, 2.0]], device="cuda")
    kw = {"disconnect_inactive_triangles": True} if disconnect else {}
    found = 0
    for order in (1, 2, 3):
        txg = [tx.clone().requires_grad_(True) for _ in range(2)]
        a = G.Scene(txg[0], rx, mesh).trace_paths(order, compact=True, **kw)
        tracer = G.ExhaustivePathTracer(**kw)
        assert not tracer.literal
        b = G.Scene(txg[1], rx, mesh).trace_paths(order, compact=True, literal=True, **kw)
        assert torch.equal(a.objects, b.objects) and torch.equal(a.keys, b.keys), order
        assert torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32))
        c = tracer.trace_rank_range(G.Scene(tx, rx, mesh), order)
        assert torch.equal(c.keys, b.keys)
        assert hasattr(tracer, "last_beam_stats")  # the whole-space call went through the pruned search
        if a.objects.shape[0]:
            for p, t in zip((a, b), txg):
                torch.sqrt((torch.diff(p.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
            assert torch.allclose(txg[0].grad, txg[1].grad, rtol=1e-5, atol=1e-6 * float(txg[1].grad.abs().max()))
        found += a.objects.shape[0]
        # a rank WINDOW is the filter kernel's business
        tracer2 = G.ExhaustivePathTracer(**kw)
        total = tracer2.num_path_candidates(G.Scene(tx, rx, mesh), order)
        w = tracer2.trace_rank_range(G.Scene(tx, rx, mesh), order, 0, total // 2)
        assert not hasattr(tracer2, "last_beam_stats") and w.objects.shape[0] <= a.objects.shape[0]
    assert found > 0


@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("with_mask", [False, True])
def test_fast_candidate_paths_equal_the_host_graph(G, canyon, assume_quads, with_mask):
    """Order 0 (every tracer) and orders 1-2 of the hybrid tracer are enumerated on the device without the host DiGraph
    (what the reference's harness runs on bruxelles.obj: orders 0 and 1, tests/benchmarks/test_rt.py:151-196): the same
    rows in the same order as DiGraph.all_paths_array over the pruned graph (_solvers.py:1013-1056, graph.rs:400-470)."""
    mesh = canyon.mesh.set_assume_quads(assume_quads)
    if with_mask:
        m = torch.ones(mesh.num_triangles, dtype=torch.bool, device="cuda")
        m[4:10] = False
        mesh = mesh.set_mask(m)
    scene = canyon.with_mesh(mesh)
    for solver in (G.HybridPathTracer(num_rays=20_000), G.ExhaustivePathTracer(disconnect_inactive_triangles=True),
                   G.ExhaustivePathTracer()):
        for order in (0, 1, 2):
            got, types = solver.generate_path_candidates(scene, order)
            graph, from_, to = solver._graph(scene)
            arr = np.asarray(graph.all_paths_array(from_, to, order + 2, include_from_and_to=False)).astype(np.int32)
            exp = arr.reshape(arr.shape[0], order) * (2 if assume_quads else 1)
            assert tuple(got.shape) == exp.shape and got.dtype == torch.int32 and tuple(types.shape) == exp.shape
            np.testing.assert_array_equal(got.cpu().numpy(), exp)
