#!/usr/bin/env python3
"""Measured support for the error model behind the beam pruning's margins.

The float32 restatement of the reference (oracle/differt_oracle.c: `intersection_of_ray_with_plane`,
_solver_image_method.py:116-135, and Moller-Trumbore, _utils.py:1263-1322) is run next to float64 on configurations
whose incidence on the mirror goes from 0 to 89.99 degrees, and the errors are split the way the argument needs:

  1. reflection point  P~ = o + t (I - o):  error ALONG the line o -> I, and LATERAL distance from that line
  2. inside test: for rays passing at a known float64 distance from a triangle edge, the largest distance at which
     the float32 decision differs from the exact one
  4. distance of P~ from the mirror plane

Expected (and printed): 1-lateral and 4 stay at a few ulp(M) whatever the incidence; 1-along grows like 1 / cos(phi);
2 stays at a few ulp(M) / sin(angle between ray and edge).  `u0` = ulp(M), M = largest coordinate magnitude.

    python oracle/studies/beam_error_model.py          (CPU only, ~20 s)
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import oracle as orc  # noqa: E402

rng = np.random.default_rng(2025)
M = 640.0
u0 = float(np.spacing(np.float32(512.0)))  # ulp of the binade of M


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def reflection_point_errors(deg, n=20000):
    """o = next reflection point, I = image (behind the mirror), plane through p with normal nrm; the segment o -> I meets
    the plane at incidence `deg`."""
    nrm = unit(rng.normal(size=(n, 3)))
    p = rng.uniform(-M, M, (n, 3))
    # a direction at the wanted angle from the normal
    t1 = unit(np.cross(nrm, rng.normal(size=(n, 3))))
    phi = np.deg2rad(deg)
    dirv = np.cos(phi) * nrm + np.sin(phi) * t1
    x = p + np.cross(nrm, t1) * rng.uniform(-50, 50, (n, 1)) + t1 * rng.uniform(-50, 50, (n, 1))  # exact crossing point
    a, b = rng.uniform(5, 300, (n, 1)), rng.uniform(5, 300, (n, 1))
    o, img = x - dirv * a, x + dirv * b
    keep = (np.abs(o).max(-1) < M) & (np.abs(img).max(-1) < 2 * M)
    o, img, p, nrm = (v[keep] for v in (o, img, p, nrm))
    o32, i32, p32, n32 = (v.astype(np.float32) for v in (o, img, p, nrm))
    got = orc.intersection_of_ray_with_plane(o32, i32 - o32, p32, n32).astype(np.float64)
    # exact crossing for the FLOAT32 INPUTS (the inputs are data; only the arithmetic is under test)
    o64, i64, p64, n64 = (v.astype(np.float64) for v in (o32, i32, p32, n32))
    d64 = i64 - o64
    t = ((p64 - o64) * n64).sum(-1) / (d64 * n64).sum(-1)
    exact = o64 + d64 * t[:, None]
    e = got - exact
    dhat = unit(d64)
    along = np.abs((e * dhat).sum(-1))
    lateral = np.linalg.norm(e - dhat * (e * dhat).sum(-1, keepdims=True), axis=-1)
    normal = np.abs(((got - p64) * unit(n64)).sum(-1))
    cosphi = np.abs((dhat * unit(n64)).sum(-1))
    return {"incidence_deg": deg, "samples": int(keep.sum()), "along_max_u0": float(along.max() / u0),
            "along_times_cos_max_u0": float((along * cosphi).max() / u0), "lateral_max_u0": float(lateral.max() / u0),
            "plane_distance_max_u0": float(normal.max() / u0)}


def inside_test_uncertainty(deg, n=200000):
    """Rays aimed to pass an edge of a triangle at a signed distance of up to +-40 u0 (float64), the ray making the
    angle `deg` with the triangle's NORMAL: the largest |distance| (x sin of the ray-edge angle) with a float32 decision
    that differs from the sign of the exact distance."""
    v0 = rng.uniform(-M / 2, M / 2, (n, 3))
    e1 = rng.normal(size=(n, 3)) * 20
    e2 = rng.normal(size=(n, 3)) * 20
    tv = np.stack([v0, v0 + e1, v0 + e2], axis=1)
    nrm = unit(np.cross(e1, e2))
    # a point on the edge v0 -> v0 + e1, displaced inside / outside by `dist` in the triangle's plane
    s = rng.uniform(0.15, 0.85, (n, 1))
    inward = unit(np.cross(nrm, e1))
    inward *= np.sign((inward * e2).sum(-1, keepdims=True))
    dist = rng.uniform(-40, 40, (n, 1)) * u0
    target = v0 + s * e1 + inward * dist
    phi = np.deg2rad(deg)
    t1 = unit(np.cross(nrm, rng.normal(size=(n, 3))))
    dirv = np.cos(phi) * nrm + np.sin(phi) * t1
    rngd = rng.uniform(20, 400, (n, 1))
    o = target - dirv * rngd
    d = dirv * rngd * rng.uniform(1.0, 1.5, (n, 1))
    keep = np.abs(o).max(-1) < M
    o, d, tv, e1k, nrmk = o[keep], d[keep], tv[keep], e1[keep], nrm[keep]
    o32, d32, tv32 = o.astype(np.float32), d.astype(np.float32), tv.astype(np.float32)
    _, hit = orc.ray_intersect_triangle(o32, d32, tv32)
    # exact signed line-to-edge-line distance for the float32 inputs (positive = inside w.r.t. that edge)
    o64, d64, tv64 = o32.astype(np.float64), d32.astype(np.float64), tv32.astype(np.float64)
    ee = tv64[:, 1] - tv64[:, 0]
    c = np.cross(d64, ee)
    cn = np.linalg.norm(c, axis=-1)
    third = tv64[:, 2] - tv64[:, 0]
    sgn = np.sign((np.cross(d64, ee) * third).sum(-1)) * np.sign((np.cross(ee, third) * d64).sum(-1))
    # side of the ray w.r.t. the edge, measured like the pyramid face test: distance between the two lines
    ld = ((tv64[:, 0] - o64) * c).sum(-1) / cn
    inside_edge = -ld * np.sign((c * np.cross(ee, third)).sum(-1) * 0 + 1.0)  # orientation fixed below by calibration
    # calibrate the orientation with clearly separated samples
    far = np.abs(ld) > 20 * u0
    flip = np.sign(np.mean((2 * hit[far] - 1) * np.sign(inside_edge[far])))
    inside_edge = inside_edge * flip
    exact = inside_edge > 0
    wrong = hit != exact
    sin_re = cn / (np.linalg.norm(d64, axis=-1) * np.linalg.norm(ee, axis=-1))
    worst = float((np.abs(ld) * sin_re)[wrong].max() / u0) if wrong.any() else 0.0
    return {"ray_to_normal_deg": deg, "samples": int(keep.sum()), "wrong_decisions": int(wrong.sum()),
            "largest_wrong_distance_times_sin_u0": worst, "largest_wrong_distance_u0": float(np.abs(ld)[wrong].max() / u0) if wrong.any() else 0.0}


def main():
    out = {"u0_m": u0, "M": M, "reflection_point": [reflection_point_errors(d) for d in (0, 30, 60, 80, 85, 88, 89, 89.9, 89.99)],
           "inside_test": [inside_test_uncertainty(d) for d in (0, 45, 70, 85, 89)]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
