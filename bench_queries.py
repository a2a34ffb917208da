"""First-hit / any-hit / visibility throughput on the 10k- and 200k-triangle synthetic Manhattan
meshes: brute-force LDS-tiled kernels vs the LBVH kernels, rays/s.

    python bench_queries.py
"""

from __future__ import annotations

import json
import time

import numpy as np


def _time(fn, reps=5):
    import torch

    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run(dev=None) -> dict:
    import torch

    import differt_amd.geometry as G
    import synthetic_scenes as S

    out = {}
    for boxes, R in ((1000, 1_000_000), (20000, 1_000_000)):
        V, Tr, _, _ = S.manhattan(boxes)
        mesh = G.Mesh(V, Tr)
        rng = np.random.default_rng(boxes)
        ext = float(np.abs(V[:, :2]).max())
        o = np.stack([rng.uniform(-ext, ext, R), rng.uniform(-ext, ext, R), rng.uniform(1, 120, R)], -1).astype(np.float32)
        tgt = np.stack([rng.uniform(-ext, ext, R), rng.uniform(-ext, ext, R), rng.uniform(1, 60, R)], -1).astype(np.float32)
        to, td = torch.as_tensor(o, device="cuda"), torch.as_tensor(tgt - o, device="cuda")
        mesh.first_triangle_hit_by_ray(to[:16], td[:16], accel="bvh")  # builds the BVH once
        T = Tr.shape[0]
        res = {"rays": R, "triangles": T}
        tb = _time(lambda: mesh.first_triangle_hit_by_ray(to, td), reps=2 if T > 50000 else 5)
        ta = _time(lambda: mesh.first_triangle_hit_by_ray(to, td, accel="bvh"))
        res["first_hit_brute_rays_per_s"] = R / tb
        res["first_hit_brute_tests_per_s"] = R * T / tb
        res["first_hit_bvh_rays_per_s"] = R / ta
        tb = _time(lambda: mesh.ray_intersect_any_triangle(to, td), reps=2 if T > 50000 else 5)
        ta = _time(lambda: mesh.ray_intersect_any_triangle(to, td, accel="bvh"))
        res["any_hit_brute_rays_per_s"] = R / tb
        res["any_hit_bvh_rays_per_s"] = R / ta
        out[f"manhattan_{T}"] = res
    return out


if __name__ == "__main__":
    print(json.dumps(run()))
