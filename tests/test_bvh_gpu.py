"""BVH-accelerated mesh queries ("next" row f1) vs the brute-force HIP operators and the oracle.

The BVH only decides WHICH triangles are tested; the test itself is the shared Moller-Trumbore, so
results must be bit-identical (indices with the reference tie-break, t, any-hit flags) whenever no
triangle test is culled wrongly.  Scenes: random soup, street canyon, 10k / 200k Manhattan, masks,
duplicated triangles (ties), axis-aligned and vertex/edge-grazing rays, single-triangle and empty
meshes.  Mirrors differt/tests/geometry/test_mesh.py:1984-2073 (Warp == pure operators).
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc
import synthetic_scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


def _check(G, mesh, o, d, batch_size=512, oracle_rows=0):
    bi, bt = mesh.first_triangle_hit_by_ray(o, d, batch_size=batch_size)
    ai, at = mesh.first_triangle_hit_by_ray(o, d, batch_size=batch_size, accel="bvh")
    assert torch.equal(ai, bi), int((ai != bi).sum())
    assert torch.equal(at.view(torch.int32), bt.view(torch.int32))
    ba = mesh.ray_intersect_any_triangle(o, d)
    aa = mesh.ray_intersect_any_triangle(o, d, accel="bvh")
    assert torch.equal(aa, ba), int((aa != ba).sum())
    if oracle_rows:
        tv = _np(mesh.triangle_vertices)
        m = None if mesh.mask is None else _np(mesh.mask)
        on, dn = (np.asarray(x, np.float32) if not isinstance(x, torch.Tensor) else _np(x) for x in (o, d))
        sel = np.random.default_rng(0).choice(len(on), oracle_rows, replace=False)
        ei, et = orc.first_triangle_hit_by_ray(on[sel], dn[sel], tv, m, batch_size=batch_size)
        np.testing.assert_array_equal(_np(ai)[sel], ei)
        np.testing.assert_array_equal(_np(at)[sel], et)
    return int((bi >= 0).sum()), int(ba.sum())


def test_random_soup(G, rng):
    T, R = 5000, 200_000
    tv = (rng.uniform(-50, 50, (T, 1, 3)) + rng.normal(size=(T, 3, 3)) * 2).astype(np.float32)
    V, Tr = tv.reshape(-1, 3), np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    mask = rng.random(T) > 0.2
    o = rng.uniform(-60, 60, (R, 3)).astype(np.float32)
    d = (rng.uniform(-60, 60, (R, 3)).astype(np.float32) - o)
    for m in (None, mask):
        nh, na = _check(G, G.Mesh(V, Tr, mask=m), o, d, oracle_rows=64)
        assert nh > 1000 and na > 1000


@pytest.mark.parametrize("batch_size", [512, 11, None])
def test_canyon_and_ties(G, rng, batch_size):
    from conftest import canyon_scene

    V, Tr = canyon_scene(rng, nextra=8)
    # duplicate a few triangles far apart in index -> exact ties across 512-tiles / BVH leaves
    Tr = np.concatenate((Tr, Tr[[0, 16, 34, 35]], Tr, Tr[[1, 17]])).astype(np.int32)
    mesh = G.Mesh(V, Tr)
    R = 100_000
    o = np.stack([rng.uniform(-25, 25, R), rng.uniform(-5, 5, R), rng.uniform(0.5, 18, R)], -1).astype(np.float32)
    d = rng.normal(size=(R, 3)).astype(np.float32) * 30
    nh, na = _check(G, mesh, o, d, batch_size=batch_size, oracle_rows=48)
    assert nh > 0.5 * R  # inside a canyon most rays hit something


def test_axis_aligned_and_grazing_rays(G):
    """Rays parallel to box faces (a == 0 exactly), through box vertices and along edges."""
    V, Tr, _, _ = S.manhattan(64)
    mesh = G.Mesh(V, Tr)
    corners = V.reshape(64, 8, 3)
    o, d = [], []
    for b in range(64):
        for c in range(8):
            p = corners[b, c]
            for dirv in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 1, 0], [1, 1, 1]):
                dv = np.asarray(dirv, np.float32) * 500
                o.append(p - dv * 0.5)          # passes exactly through the corner
                d.append(dv)
                o.append(p + np.float32(1e-3))  # starts next to the corner
                d.append(dv)
    o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
    _check(G, mesh, o, d, oracle_rows=64)
    # rays between box corners (edge-aligned segments, un-normalised, like TX->RX segments)
    rng = np.random.default_rng(9)
    a = V[rng.integers(0, len(V), 50_000)]
    b = V[rng.integers(0, len(V), 50_000)]
    _check(G, mesh, a, (b - a).astype(np.float32), oracle_rows=64)


@pytest.mark.parametrize("boxes", [1000, 20000])
def test_manhattan(G, boxes):
    """10k and 200k triangles (configs[2] / configs[4] meshes): 1e6 rays."""
    V, Tr, centres, heights = S.manhattan(boxes)
    mesh = G.Mesh(V, Tr)
    rng = np.random.default_rng(boxes)
    R = 1_000_000 if boxes == 1000 else 200_000
    ext = np.abs(V[:, :2]).max()
    o = np.stack([rng.uniform(-ext, ext, R), rng.uniform(-ext, ext, R), rng.uniform(1, 120, R)], -1).astype(np.float32)
    tgt = np.stack([rng.uniform(-ext, ext, R), rng.uniform(-ext, ext, R), rng.uniform(1, 60, R)], -1).astype(np.float32)
    nh, na = _check(G, mesh, o, (tgt - o).astype(np.float32), oracle_rows=24)
    assert nh > 0.3 * R


def test_tiny_meshes(G):
    o = np.array([[0.25, 0.25, 1.0], [5, 5, 1.0]], np.float32)
    d = np.array([[0, 0, -2.0], [0, 0, -2.0]], np.float32)
    one = G.Mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.int32))
    _check(G, one, o, d)
    i, t = one.first_triangle_hit_by_ray(o, d, accel="bvh")
    assert _np(i).tolist() == [0, -1] and _np(t)[0] == 0.5
    two = G.Mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, -1], [1, 0, -1], [0, 1, -1]], np.float32),
                 np.array([[0, 1, 2], [3, 4, 5]], np.int32))
    _check(G, two, o, d)
    e = G.Mesh.empty()
    i, t = e.first_triangle_hit_by_ray(o, d, accel="bvh")
    assert (_np(i) == -1).all() and np.isinf(_np(t)).all()
    assert not _np(e.ray_intersect_any_triangle(o, d, accel="bvh")).any()


@pytest.mark.parametrize(
    "case", ["far_from_origin", "flat", "huge_extent", "non_finite", "tiny", "big_far"], ids=str
)
def test_quantised_wide_nodes_stay_conservative(G, rng, case):
    """The 4-ary nodes hold their child boxes on a 16-bit grid over the scene (csrc/bvh.hpp, Bvh4Node), rounded
    outward with the decode expression itself: results stay bit-identical to the brute-force operators where the
    grid is coarser than the coordinates' float spacing is fine (far from the origin), degenerate (flat scene),
    stretched by one remote triangle, or unusable (a non-finite vertex: binary walk)."""
    T, R = 6000, 30000
    tv = rng.uniform(-30, 30, (T, 3, 3)).astype(np.float32)
    tv[:, 1:] = tv[:, :1] + rng.normal(0, 1.0, (T, 2, 3)).astype(np.float32)
    shift = np.zeros(3, np.float32)
    if case == "far_from_origin":  # ulp(1e6) = 0.0625 > the grid cell (60 / 65535)
        shift = np.array([1.0e6, -2.0e6, 5.0e5], np.float32)
    elif case == "big_far":  # ordered walks on the wide tree need >= 65536 triangles
        T = 70000
        tv = rng.uniform(-300, 300, (T, 3, 3)).astype(np.float32)
        tv[:, 1:] = tv[:, :1] + rng.normal(0, 2.0, (T, 2, 3)).astype(np.float32)
        shift = np.array([3.0e5, 3.0e5, -3.0e5], np.float32)
    elif case == "flat":
        tv[..., 2] = 0.0
    elif case == "huge_extent":
        tv[0] = np.array([[1e30, 0, 0], [1e30, 1e28, 0], [1e30, 0, 1e28]], np.float32)
    elif case == "non_finite":
        tv[0, 1, 0] = np.inf
        tv[1, 2, 2] = np.nan
        tv[2, 0, 1] = -np.inf
    elif case == "tiny":
        tv *= 1e-20
    tv = tv + shift
    span = 40.0 if case != "big_far" else 350.0
    scale = 1e-20 if case == "tiny" else 1.0
    o = (rng.uniform(-span, span, (R, 3)) * scale).astype(np.float32) + shift
    tgt = tv[rng.integers(0, T, R), rng.integers(0, 3, R)]  # aim at vertices: grazing box faces
    tgt = np.where(np.isfinite(tgt), tgt, 0.0).astype(np.float32)
    d = (tgt - o) * rng.uniform(0.5, 2.0, (R, 1)).astype(np.float32)
    d[: R // 4] = rng.normal(size=(R // 4, 3)).astype(np.float32) * np.float32(scale * 10)
    mesh = G.Mesh(tv.reshape(-1, 3), np.arange(3 * T, dtype=np.int32).reshape(T, 3))
    hits, _ = _check(G, mesh, o, d)
    if case in ("far_from_origin", "big_far", "flat"):
        assert hits > R // 100


def test_bvh_first_hit_gradients(G):
    """The BVH path keeps the custom VJP (test_mesh.py:2028-2072)."""
    mesh0 = G.Mesh.box(2.0, 2.0, 2.0)
    o = torch.tensor([[0.0, 0.0, 3.0], [0.0, 3.0, 0.0], [3.0, 0.0, 0.0]], device="cuda", requires_grad=True)
    d = torch.tensor([[0.0, 0.0, -1.0], [0.0, -1.0, 0.0], [-1.0, 0.0, 0.0]], device="cuda", requires_grad=True)
    v = mesh0.vertices.clone().requires_grad_(True)
    grads = []
    for accel in (None, "bvh"):
        for x in (o, d, v):
            x.grad = None
        _, t = G.Mesh(v, mesh0.triangles).first_triangle_hit_by_ray(o, d, accel=accel)
        t.sum().backward()
        grads.append([x.grad.clone() for x in (o, d, v)])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_trace_with_bvh_occlusion_equals_brute_force(G, rng, order, assume_quads):
    """Tracer with `accel="bvh"` (occlusion stage on the LBVH): dense mask and compact output identical
    to the default brute-force stage and to the oracle."""
    from conftest import canyon_case

    if order == 0:
        V, Tr, mask, tx, rx, _ = canyon_case(rng, 1, assume_quads)
        cand = np.zeros((1, 0), np.int32)
    else:
        V, Tr, mask, tx, rx, cand = canyon_case(rng, order, assume_quads)
    scene = G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads))
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand, mask=mask, assume_quads=assume_quads)
    fast = scene.trace_paths(path_candidates=cand, solver=G.ExhaustivePathTracer(accel="bvh"))
    np.testing.assert_array_equal(_np(fast.mask), o["mask"])
    np.testing.assert_array_equal(_np(fast.vertices).view(np.uint32), o["vertices"].view(np.uint32))
    cp = scene.trace_paths(path_candidates=cand, solver=G.ExhaustivePathTracer(accel="bvh"), compact=True)
    np.testing.assert_array_equal(_np(cp.keys), np.flatnonzero(o["mask"].reshape(-1)))


def test_cfg3_window_bvh_vs_brute(G):
    """configs[2] scene, 5e6 ranks x 1024 pairs: identical valid keys with either occlusion stage."""
    V, Tr, centres, heights = S.manhattan(1000)
    tx, rx = S.manhattan_tx_rx(centres, heights, 16, 64)
    scene = G.Scene(tx, rx, G.Mesh(V, Tr))
    lo, hi = 40_000_000, 45_000_000
    a = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, lo, hi, max_survivors=1 << 22)
    b = G.ExhaustivePathTracer(accel="bvh").trace_rank_range_literal(scene, 2, lo, hi, max_survivors=1 << 22)
    assert torch.equal(a.keys, b.keys) and torch.equal(a.vertices, b.vertices)


def test_bvh_visibility_equals_brute_force(G):
    """Mesh.triangles_visible_from_vertex(accel="bvh") == the LDS-tiled kernel (same lattice, same
    first-hit rule) on cube / masked box-in-box / Manhattan."""
    Vc, Tc = orc.box_mesh(with_top=True)
    cube = G.Mesh(Vc, Tc)
    for p in ([2.0, 0, 0], [2.0, 2.0, 0], [2.0, 2.0, 2.0]):
        a = cube.triangles_visible_from_vertex(np.asarray(p, np.float32), num_rays=10_000)
        b = cube.triangles_visible_from_vertex(np.asarray(p, np.float32), num_rays=10_000, accel="bvh")
        assert torch.equal(a, b)
    V, Tr, centres, heights = S.manhattan(1000)
    tx, rx = S.manhattan_tx_rx(centres, heights, 2, 3)
    mask = np.random.default_rng(0).random(Tr.shape[0]) > 0.1
    for m in (None, mask):
        mesh = G.Mesh(V, Tr, mask=m)
        pts = np.concatenate((tx, rx))
        a = mesh.triangles_visible_from_vertex(pts, num_rays=200_000)
        b = mesh.triangles_visible_from_vertex(pts, num_rays=200_000, accel="bvh")
        assert torch.equal(a, b) and int(a.sum()) > 50


def test_overflow_path_of_the_lds_stack(tmp_path):
    """The traversal stack is a 20-entry LDS column per lane; a walk that needs more tests the far subtree at once
    through the node's leaf range.  A library built with a 2-entry column takes that path on almost every walk:
    this whole file must pass against it unchanged (bit-identical to brute force and to the oracle)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    lib = tmp_path / "libdiffert_amd_stack2.so"
    env = {**os.environ, "DIFFERT_AMD_LIB": str(lib), "DRT_EXTRA_FLAGS": "-DDRT_BVH_LDS_STACK_N=2"}
    r = subprocess.run([sys.executable, "-m", "differt_amd.build"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0 and lib.exists(), r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_bvh_gpu.py", "-m", "gpu", "-q", "-x", "-k",
                        "not overflow_path"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
