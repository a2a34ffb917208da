"""First-hit / any-hit / visibility throughput on the 10k- and 200k-triangle synthetic Manhattan
meshes: brute-force LDS-tiled kernels vs the LBVH kernels, rays/s; on the 10k mesh
also the ray-launching users of the LBVH (rows f2 / f3): SBR path launching and the visibility estimate.

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
        if T <= 50000:  # rows f2 / f3: ray-launching users of the LBVH first hit
            _, _, c, h = S.manhattan(boxes)
            tx1, rx4 = S.manhattan_tx_rx(c, h, 1, 4)
            scene = G.Scene(torch.as_tensor(tx1, device="cuda"), torch.as_tensor(rx4, device="cuda"), mesh)
            sbr = G.SBRPathLauncher(num_rays=1_000_000, max_dist=1.0)
            for order in (1, 3):
                ts = _time(lambda: sbr.launch_paths(scene, order), reps=3)
                res[f"sbr_order{order}_rays_per_s"] = 1_000_000 / ts
                res[f"sbr_order{order}_bounces_per_s"] = 1_000_000 * order / ts
            res["sbr_note"] = "SBRPathLauncher.launch_paths, 1 TX x 4 RX, 1e6 lattice rays, LBVH first hit per bounce"
            tv = _time(lambda: mesh.triangles_visible_from_vertex(scene.transmitters.reshape(-1, 3), num_rays=1_000_000,
                                                                  accel="bvh"), reps=3)
            res["visibility_1e6_rays_s"] = tv
        out[f"manhattan_{T}"] = res
    return out


if __name__ == "__main__":
    print(json.dumps(run()))
