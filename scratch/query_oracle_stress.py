"""GPU any-hit / first-hit (brute-force LDS-tiled kernels, shared and per-ray triangle sets, masks,
tile sizes, duplicated triangles) vs the CPU oracle.  python scratch/query_oracle_stress.py [seconds]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(5)
st = {"cases": 0, "tests": 0, "any_mismatch": 0, "idx_mismatch": 0, "t_mismatch": 0, "hits": 0}
t0 = time.time()
while time.time() - t0 < budget:
    per_ray = rng.random() < 0.2
    R = int(rng.integers(1, 3000 if not per_ray else 200))
    T = int(rng.integers(0, 6000 if not per_ray else 300))
    scale = np.float32(10.0 ** rng.uniform(-2, 3))
    tshape = (R, T, 3, 3) if per_ray else (T, 3, 3)
    tv = (rng.normal(size=tshape) * 2).astype(np.float32) * scale
    if T > 8 and not per_ray:  # exact duplicates -> ties
        src = rng.integers(0, T, 6)
        tv[rng.integers(0, T, 6)] = tv[src]
    o = (rng.normal(size=(R, 3)) * 4).astype(np.float32) * scale
    if T and not per_ray:
        d = (tv.mean(axis=1)[rng.integers(0, T, R)] - o).astype(np.float32) * np.float32(rng.choice([1.0, 1.5, 3.0]))
    else:
        d = (rng.normal(size=(R, 3)) * 6).astype(np.float32) * scale
    act = None
    if rng.random() < 0.5:
        act = rng.random((R, T) if (per_ray and rng.random() < 0.5) else (T,)) > 0.3
    bs = rng.choice([None, 512, 11, 1, 100])
    bs = None if bs is None else int(bs)
    eps = None if rng.random() < 0.7 else float(10.0 ** rng.uniform(-7, -2))
    tol = None if rng.random() < 0.7 else float(rng.choice([0.0, 1e-3, -0.5, 0.5]))
    ea = orc.ray_intersect_any_triangle(o, d, tv, act, epsilon=eps, hit_tol=tol)
    ei, et = orc.first_triangle_hit_by_ray(o, d, tv, act, batch_size=bs, epsilon=eps)
    ga = G.ray_intersect_any_triangle(o, d, tv, act, epsilon=eps, hit_tol=tol).cpu().numpy()
    gi, gt = G.first_triangle_hit_by_ray(o, d, tv, act, batch_size=bs, epsilon=eps)
    gi, gt = gi.cpu().numpy(), gt.cpu().numpy()
    st["cases"] += 1
    st["tests"] += R * T
    st["hits"] += int((ei >= 0).sum())
    st["any_mismatch"] += int((ga != ea).sum())
    st["idx_mismatch"] += int((gi != ei).sum())
    st["t_mismatch"] += int((gt.view(np.uint32) != et.view(np.uint32)).sum())
st["seconds"] = time.time() - t0
print(json.dumps(st))
