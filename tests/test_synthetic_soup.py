"""The triangle-soup generator of the stress drivers (synthetic_scenes.soup_city, round 6): it must really produce what
VERDICT r05 found missing from every generated scene -- ear-clipped, gable and hip roofs, slivers of aspect >= 1e3,
T-junctions, duplicated vertices, full 3-D rotations, coordinates 1e4-1e5 m from the origin -- and be reproducible."""

from __future__ import annotations

import numpy as np

import synthetic_scenes as S


def _poly_area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def test_ear_clipping_covers_the_polygon():
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(3, 10))
        ang = (np.arange(n) + rng.uniform(-0.2, 0.2, n)) * (2 * np.pi / n) + rng.uniform(0, 2 * np.pi)
        poly = np.column_stack((np.cos(ang), np.sin(ang))) * rng.uniform(0.4, 1.0, (len(ang), 1)) * 10
        tris = S._ear_clip(poly)
        assert len(tris) == len(poly) - 2
        areas = [_poly_area(poly[list(t)]) for t in tris]
        assert min(areas) > 0  # every ear keeps the polygon's orientation
        assert abs(sum(areas) - _poly_area(poly)) < 1e-9 * max(1.0, _poly_area(poly))


def test_soups_contain_the_features_they_promise():
    rng = np.random.default_rng(11)
    tot: dict = {}
    slivers = flat = far = 0
    planes_axis_aligned = 0
    for _ in range(60):
        V, Tr, info = S.soup_city(rng, int(rng.integers(2, 20)))
        assert V.dtype == np.float32 and Tr.dtype == np.int32 and Tr.min() >= 0 and Tr.max() < len(V)
        assert info["triangles"] == len(Tr)
        for k, v in info.items():
            if k != "offset_m":
                tot[k] = tot.get(k, 0) + v
        far += info["offset_m"] >= 1e4
        tv = V[Tr].astype(np.float64)
        e = np.stack([np.linalg.norm(tv[:, (k + 1) % 3] - tv[:, k], axis=1) for k in range(3)], 1)
        nrm = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0])
        area = 0.5 * np.linalg.norm(nrm, axis=1)
        slivers += int((e.max(1) ** 2 / np.maximum(2 * area, 1e-300) >= 1e3).sum())
        flat += int((area == 0).sum())
        u = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
        planes_axis_aligned += int((np.abs(u).max(axis=1) > 1 - 1e-9).sum())
        if info["offset_m"]:
            assert np.abs(V).max() > 0.5 * info["offset_m"]
        # duplicated positions under different indices
        if info["duplicated"]:
            assert len(np.unique(V, axis=0)) < len(V)
    for k in ("gable", "hip", "nonconvex", "sliver_walls", "sliver_ears", "t_junctions", "duplicated", "thin_edge"):
        assert tot[k] > 0, (k, tot)
    assert slivers > 50 and far > 5
    assert planes_axis_aligned == 0  # a uniform 3-D rotation leaves no triangle in a coordinate plane


def test_soup_is_reproducible_and_end_points_cover_the_scene():
    a = S.soup_city(np.random.default_rng(3), 7)
    b = S.soup_city(np.random.default_rng(3), 7)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    rng = np.random.default_rng(4)
    tx, rx = S.soup_end_points(rng, a[0], 3, 50)
    assert tx.shape == (3, 3) and rx.shape == (50, 3) and tx.dtype == np.float32
    lo, hi = a[0].min(0), a[0].max(0)
    c, e = (lo + hi) / 2, (hi - lo) / 2
    assert (np.abs(rx - c) <= 1.11 * e + 1e-3 * np.abs(c).max()).all()
