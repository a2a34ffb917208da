"""The one reference known answer the visibility oracle does not reproduce
(differt/tests/geometry/test_utils.py:770-827, unmasked box-in-box seen from RX: 12 in the reference,
11 here): how far is the 12th triangle from being counted?

For every triangle that no lattice ray reaches FIRST, find the ray that comes closest to it -- the
smallest violation of Moller-Trumbore's barycentric conditions among the rays whose plane intersection
is the nearest one -- and report that violation next to float32's resolution.  If the margin is a few
ulps of the barycentric coordinates, the count depends on the last bits of cos / arccos / sin in the
lattice construction (XLA vs NumPy), i.e. the known answer is a property of one libm, not of the
algorithm.

    python oracle/studies/visibility_margin.py  ->  JSON (committed as profiles/r02/visibility_margin.json)
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import oracle as orc  # noqa: E402

Vo, To = orc.box_mesh(4.0, 4.0, 4.0)
Vi, Ti = orc.box_mesh(1.0, 1.0, 1.0)
V, Tr = np.concatenate((Vo, Vi)), np.concatenate((To, Ti + 8))
tv = orc.triangle_vertices(V, Tr)
out = {}
for name, vertex in (("tx", [-1.0, 0.0, 0.0]), ("rx", [1.0, 0.0, 0.0])):
    v = np.asarray(vertex, np.float32)
    n = 1_000_000
    centers = tv.mean(axis=-2, keepdims=True, dtype=np.float32)
    world = np.concatenate((tv, centers), axis=-2).reshape(-1, 3)
    fr = orc.viewing_frustum(v, world)
    dirs = orc.fibonacci_lattice(n, frustum=fr)
    idx, t = orc.first_triangle_hit_by_ray(np.broadcast_to(v, dirs.shape), dirs, tv, None, batch_size=None)
    seen = np.zeros(len(tv), bool)
    seen[idx[idx >= 0]] = True
    # float64 Moller-Trumbore of every ray against every unseen triangle: barycentric violation
    d64, o64 = dirs.astype(np.float64), v.astype(np.float64)
    rows = []
    for j in np.flatnonzero(~seen):
        v0, v1, v2 = tv[j].astype(np.float64)
        e1, e2 = v1 - v0, v2 - v0
        h = np.cross(d64, e2)
        a = h @ e1
        ok = np.abs(a) > 1e-12
        f = np.where(ok, 1.0 / np.where(ok, a, 1.0), 0.0)
        s = o64 - v0
        u = f * (h @ s)
        q = np.cross(s, e1)
        vv = f * (d64 @ q)
        tt = f * (q @ e2)
        viol = np.maximum.reduce([-u, -vv, u + vv - 1.0, np.zeros_like(u)])  # 0 = inside
        cand = ok & (tt > 1e-6)
        # only rays for which this triangle would be the FIRST hit matter
        nearer = cand & (tt <= np.where(idx >= 0, t.astype(np.float64), np.inf) * (1 + 1e-6))
        if nearer.any():
            k = int(np.argmin(np.where(nearer, viol, np.inf)))
            rows.append({"triangle": int(j), "closest_ray": k, "barycentric_violation": float(viol[k]),
                         "violation_in_float32_ulps_of_1": float(viol[k] / np.finfo(np.float32).eps),
                         "ray_is_on_frustum_boundary": bool(k in (0, n - 1) or
                                                            abs(float(np.arccos(np.clip(dirs[k, 2], -1, 1))) - float(fr[0, 1])) < 1e-6
                                                            or abs(float(np.arccos(np.clip(dirs[k, 2], -1, 1))) - float(fr[1, 1])) < 1e-6)})
    rows.sort(key=lambda r: r["barycentric_violation"])
    out[name] = {"visible_float32_oracle": int(seen.sum()), "reference_known_answer": 11 if name == "tx" else 12,
                 "nearest_unseen_triangles": rows[:3]}
print(json.dumps(out, indent=1))
