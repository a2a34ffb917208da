"""The reference's own in-tree meshes (docs/source/notebooks/{bruxelles,manhattan,manhattan_small}.obj, read by the rule of
differt-core/src/geometry/mesh.rs:399-429 into tests/golden/*.npz by tests/golden/make_golden.py; bruxelles.obj is the
mesh of the reference's benchmark harness, differt/tests/benchmarks/fixtures.py:43-68, test_rt.py:77-196).

Every earlier scene-level parity test ran on axis-aligned box cities.  These meshes are triangle soups: walls at any
angle, roofs that are ear-clipped polygons, triangles in no particular order, duplicated vertex positions.
  (a) dense + compact tracer == the C oracle, bit for bit, on candidate windows around real valid paths;
  (b) the pruned search == the exhaustive tracer over WHOLE candidate spaces (order 2 on all three, order 3 on
      manhattan_small), kappa 64 and 1, clustered and plain mappings, pairing pass on and off;
  (c) LBVH any / first hit == brute force == oracle on the harness's 10 000 lattice rays with a 50 % mask;
  (d) the pairing pass itself: what it pairs IS the same mirror, and it finds thousands of pairs where the (2i, 2i+1)
      rule of round 4 found 36.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

import oracle as orc
import synthetic_scenes as S

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"
MESHES = ("bruxelles", "manhattan", "manhattan_small")


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def load(name):
    return S.load_real_mesh(name)


def _np(t):
    return t.detach().cpu().numpy()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def end_points(G, V, Tr, ntx, nrx, seed=7):
    return S.outdoor_end_points(G, V, Tr, ntx, nrx, seed)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", MESHES)
def test_pairing_pass_pairs_the_same_mirror(G, name):
    V, Tr = load(name)
    mesh = G.Mesh(V, Tr)
    info = mesh.beam_pairing()
    T = Tr.shape[0]
    assert info["pair_mode"] and info["primitives"] == T - info["pairs"]
    # bruxelles: 5 829 pairs by the CPU study (scratch/pairing_study.py); the (2i, 2i+1) rule of round 4 found 36
    assert info["pairs"] >= {"bruxelles": 5500, "manhattan": 1000, "manhattan_small": 850}[name], info
    # the table itself, through the debug read-back of the rows: every pair is (v0 v1 v2), (v0 v2 v3) with == normals
    from differt_amd import _lib
    import ctypes as C

    prims, pairs = C.c_int64(), C.c_int64()
    assert _lib.load().drt_mesh_beam_pairing(mesh.handle().h, C.byref(prims), C.byref(pairs)) == 1
    table = torch.empty((prims.value, 2), dtype=torch.int32, device="cuda")
    _lib.call("drt_mesh_beam_pairing_table", mesh.handle().h, table.data_ptr(), prims.value, None)
    torch.cuda.synchronize()
    tab = _np(table)
    assert (tab[:, 0] >= 0).all() and ((tab[:, 1] >= 0).sum() == pairs.value)
    used = np.concatenate([tab[:, 0], tab[tab[:, 1] >= 0, 1]])
    assert np.array_equal(np.sort(used), np.arange(T)), "every triangle in exactly one primitive"
    tv = V[Tr]
    nrm = _np(mesh.handle().normals())
    a, b = tab[tab[:, 1] >= 0, 0], tab[tab[:, 1] >= 0, 1]
    assert np.array_equal(tv[a][:, 0], tv[b][:, 0]) and np.array_equal(tv[a][:, 2], tv[b][:, 1])
    assert (nrm[a] == nrm[b]).all()


@pytest.mark.parametrize("name", MESHES)
def test_tracer_vs_oracle_around_real_paths(G, name):
    """(a) Candidate tables built around the valid order-1 / order-2 paths of the scene (found by the pruned search),
    plus random rows: dense and compact tracer == the C oracle (mask, objects, vertex bits)."""
    V, Tr = load(name)
    tx, rx = end_points(G, V, Tr, 6, 40)
    mesh = G.Mesh(V, Tr)
    scene = G.Scene(tx, rx, mesh)
    tracer = G.ExhaustivePathTracer()
    rng = np.random.default_rng(3)
    T = Tr.shape[0]
    nvalid = 0
    for order in (1, 2):
        found = _np(tracer.trace_beam_pruned(scene, order, max_paths=1 << 16).objects)[:, 1:-1]
        rows = [found]
        if len(found):
            for _ in range(6):  # neighbours of valid rows: one mirror replaced
                r = found.copy()
                r[np.arange(len(r)), rng.integers(0, order, len(r))] = rng.integers(0, T, len(r))
                rows.append(r)
        rows.append(rng.integers(0, T, (1500, order)))
        cand = np.unique(np.concatenate(rows).astype(np.int32), axis=0)
        if order > 1:
            cand = cand[(np.diff(cand, axis=1) != 0).all(1)]  # rows of the reference's graph: no immediate repeat
        cand = cand[:4000]
        o = orc.trace_path_candidates(V, Tr, tx, rx, cand)
        got = scene.trace_paths(path_candidates=cand)
        np.testing.assert_array_equal(_np(got.mask), o["mask"])
        np.testing.assert_array_equal(_np(got.objects), o["objects"])
        np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(o["vertices"]))
        comp = tracer.trace_path_candidates_compact(scene, cand)
        m = o["mask"].reshape(-1)
        np.testing.assert_array_equal(_np(comp.keys), np.flatnonzero(m))
        np.testing.assert_array_equal(_bits(_np(comp.vertices)), _bits(o["vertices"].reshape(-1, order + 2, 3)[m]))
        nvalid += int(m.sum())
    assert nvalid >= 40, nvalid


def _same(a, b, what):
    assert a.objects.shape == b.objects.shape, (what, tuple(a.objects.shape), tuple(b.objects.shape))
    assert torch.equal(a.objects, b.objects), what
    assert torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32)), what


@pytest.mark.parametrize("name", MESHES)
def test_pruned_equals_exhaustive_whole_order2_space(G, name):
    """(b) 4 TX x 16 RX, ALL n (n-1) order-2 candidates of every pair (2.0e8 per pair on bruxelles) through the
    exhaustive tracer == the pruned search: pairing pass on / off, clustered / plain expansion, kappa 64 and 1."""
    V, Tr = load(name)
    tx, rx = end_points(G, V, Tr, 4, 16)
    scene = G.Scene(tx, rx, G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer(accel="bvh")
    total = 0
    for order in (1, 2):
        ex = tracer.trace_rank_range_literal(scene, order, max_survivors=1 << 25, max_paths=1 << 20)
        for kappa in (64.0, 1.0):
            for pairs in (True, False):
                for expansion in ("auto", "plain", "fused") if kappa == 64.0 else ("auto",):
                    bp = tracer.trace_beam_pruned(scene, order, kappa=kappa, pairs=pairs, expansion=expansion, max_paths=1 << 18)
                    _same(bp, ex, (name, order, kappa, pairs, expansion, tracer.last_beam_stats))
                    assert tracer.last_beam_stats["pair_mode"] == pairs
        # the row-by-row trace of the pair rows and the clustered receiver stage as further mappings
        _same(tracer.trace_beam_pruned(scene, order, rows="plain", max_paths=1 << 18), ex, (name, order, "rows"))
        _same(tracer.trace_beam_pruned(scene, order, emit="clustered", max_paths=1 << 18), ex, (name, order, "emit"))
        total += ex.objects.shape[0]
    assert total >= 40, total


def test_pruned_equals_exhaustive_whole_order3_spaces_manhattan_small(G):
    """(b) order 3 on manhattan_small: the WHOLE 1.35e10-candidate space of 12 (tx, rx) pairs through the exhaustive
    tracer == the pruned search's rows of those pairs."""
    V, Tr = load("manhattan_small")
    tx, rx = end_points(G, V, Tr, 2, 6)
    scene = G.Scene(tx, rx, G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer(accel="bvh")
    ex = tracer.trace_rank_range_literal(scene, 3, max_survivors=1 << 25, max_paths=1 << 20)
    for kappa in (64.0, 1.0):
        for pairs in (True, False):
            bp = tracer.trace_beam_pruned(scene, 3, kappa=kappa, pairs=pairs, max_paths=1 << 18)
            _same(bp, ex, (kappa, pairs, tracer.last_beam_stats))
    _same(tracer.trace_beam_pruned(scene, 3, expansion="plain", max_paths=1 << 18), ex, "plain")
    _same(tracer.trace_beam_pruned(scene, 3, rows="plain", max_paths=1 << 18), ex, "rows")
    assert ex.objects.shape[0] >= 1


@pytest.mark.parametrize("name", MESHES)
def test_queries_on_the_harness_rays(G, name):
    """(c) reference harness shapes (test_rt.py:77-147): 10 000 Fibonacci-lattice rays from a transmitter 10 m above
    the mesh centre, a random 50 % triangle mask: LBVH == brute force == oracle for any-hit and first-hit."""
    V, Tr = load(name)
    rng = np.random.default_rng(11)
    mask = rng.random(Tr.shape[0]) < 0.5
    centre = V.mean(0)
    origin = (centre + np.array([0, 0, 10.0], np.float32)).astype(np.float32)
    lattice = _np(G.fibonacci_lattice(10_000)).astype(np.float32)
    mesh = G.Mesh(V, Tr, mask=mask)
    tv = V[Tr]
    # the harness's literal call (unit directions: the any-hit operator tests the SEGMENT o -> o + d, so nearly nothing
    # is blocked) and the same lattice as 1 km segments
    for scale in (1.0, 1000.0):
        d = (lattice * np.float32(scale)).astype(np.float32)
        o = np.broadcast_to(origin, d.shape).copy()
        e_any = orc.ray_intersect_any_triangle(o, d, tv, active_triangles=mask)
        e_idx, e_t = orc.first_triangle_hit_by_ray(o, d, tv, active_triangles=mask)
        for accel in (None, "bvh"):
            got = mesh.ray_intersect_any_triangle(o, d, accel=accel)
            np.testing.assert_array_equal(_np(got), e_any)
            idx, t = mesh.first_triangle_hit_by_ray(o, d, accel=accel)
            np.testing.assert_array_equal(_np(idx), e_idx)
            np.testing.assert_array_equal(_bits(_np(t)), _bits(e_t))
        assert 0 < (e_idx >= 0).sum() < e_idx.size
    assert 0 < e_any.sum() < e_any.size  # the long segments
