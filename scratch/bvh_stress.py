"""Stress comparison of the LBVH queries against the brute-force kernels (the normative operators):
counts mismatching any-hit flags, first-hit indices and t bit patterns over many scenes and rays,
including segment-style rays between mesh vertices (exactly grazing edges / coplanar faces).

    python scratch/bvh_stress.py [total_rays_in_millions]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

target = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 100e6
rng = np.random.default_rng(2024)
stats = {"rays": 0, "any_mismatch": 0, "idx_mismatch": 0, "t_mismatch": 0, "scenes": 0, "hits": 0}
t0 = time.time()
while stats["rays"] < target:
    kind = stats["scenes"] % 4
    if kind == 0:
        V, Tr, _, _ = S.manhattan(int(rng.integers(50, 3000)), seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        T = int(rng.integers(100, 20000))
        tv = (rng.uniform(-100, 100, (T, 1, 3)) + rng.normal(size=(T, 3, 3)) * rng.uniform(0.1, 20)).astype(np.float32)
        V, Tr = tv.reshape(-1, 3), np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    elif kind == 2:  # coplanar grid of quads (many exact ties / edge hits)
        n = int(rng.integers(4, 60))
        xs = np.arange(n + 1, dtype=np.float32) * np.float32(rng.choice([0.5, 1.0, 3.0]))
        gx, gy = np.meshgrid(xs, xs, indexing="ij")
        V = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size, np.float32)], -1).astype(np.float32)
        idx = lambda i, j: i * (n + 1) + j  # noqa: E731
        Tr = np.asarray([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(n) for j in range(n)] +
                        [[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(n) for j in range(n)], np.int32)
    else:
        V, Tr, _, _ = S.manhattan(int(rng.integers(10, 400)), pitch=float(rng.uniform(25, 60)), seed=int(rng.integers(1 << 30)))
    mask = (rng.random(Tr.shape[0]) > 0.1) if rng.random() < 0.5 else None
    mesh = G.Mesh(V, Tr, mask=mask)
    R = 2_000_000
    lo, hi = V.min(0) - 5, V.max(0) + 5
    mode = int(rng.integers(0, 3))
    if mode == 0:  # random segments through the scene
        o = rng.uniform(lo, hi, (R, 3)).astype(np.float32)
        d = (rng.uniform(lo, hi, (R, 3)).astype(np.float32) - o)
    elif mode == 1:  # vertex-to-vertex segments (grazing edges, lying in faces)
        o = V[rng.integers(0, len(V), R)]
        d = (V[rng.integers(0, len(V), R)] - o).astype(np.float32)
    else:  # axis-aligned rays from jittered vertices
        o = (V[rng.integers(0, len(V), R)] + rng.normal(size=(R, 3)).astype(np.float32) * np.float32(rng.choice([0, 1e-3, 1.0]))).astype(np.float32)
        d = np.eye(3, dtype=np.float32)[rng.integers(0, 3, R)] * rng.choice([-1.0, 1.0], (R, 1)).astype(np.float32) * 300
    to, td = torch.as_tensor(o, device="cuda"), torch.as_tensor(d, device="cuda")
    bs = int(rng.choice([512, 11, 0]))
    bi, bt = mesh.first_triangle_hit_by_ray(to, td, batch_size=bs or None)
    ai, at = mesh.first_triangle_hit_by_ray(to, td, batch_size=bs or None, accel="bvh")
    ba = mesh.ray_intersect_any_triangle(to, td)
    aa = mesh.ray_intersect_any_triangle(to, td, accel="bvh")
    stats["rays"] += R
    stats["scenes"] += 1
    stats["hits"] += int((bi >= 0).sum())
    stats["any_mismatch"] += int((aa != ba).sum())
    stats["idx_mismatch"] += int((ai != bi).sum())
    stats["t_mismatch"] += int((at.view(torch.int32) != bt.view(torch.int32)).sum())
stats["seconds"] = time.time() - t0
print(json.dumps(stats))
