"""GPU dense Moller-Trumbore vs the CPU oracle, bit for bit, over many input distributions
(random, structured axis-aligned boxes, degenerate triangles, huge / tiny scales, rays in triangle
planes).  Uses every host core for the oracle.

    python scratch/oracle_stress.py [seconds]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import oracle as orc  # noqa: E402
import synthetic_scenes as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(99)
st = {"cases": 0, "tests": 0, "hits": 0, "hit_mismatch": 0, "t_mismatch": 0}
t0 = time.time()
while time.time() - t0 < budget:
    kind = st["cases"] % 5
    R, T = int(rng.integers(64, 2048)), int(rng.integers(16, 8192))
    scale = np.float32(10.0 ** rng.uniform(-3, 4))
    if kind == 0:
        tv = (rng.uniform(-1, 1, (T, 1, 3)) * 50 + rng.normal(size=(T, 3, 3)) * 2).astype(np.float32) * scale
        o = (rng.uniform(-1, 1, (R, 3)) * 50).astype(np.float32) * scale
        d = (rng.uniform(-1, 1, (R, 3)) * 50).astype(np.float32) * scale - o
    elif kind == 1:  # axis-aligned boxes, rays between box vertices
        V, Tr, _, _ = S.manhattan(int(rng.integers(4, 400)), seed=int(rng.integers(1 << 30)))
        tv = orc.triangle_vertices(V, Tr)[:T]
        o = V[rng.integers(0, len(V), R)]
        d = (V[rng.integers(0, len(V), R)] - o).astype(np.float32)
    elif kind == 2:  # rays aimed at points inside triangles (many hits, edge cases on u+v==1)
        tv = (rng.normal(size=(T, 3, 3)) * 3).astype(np.float32) * scale
        w = rng.dirichlet(np.ones(3), R).astype(np.float32)
        w[: R // 4] = np.round(w[: R // 4] * 2) / 2  # vertices / edge midpoints exactly
        tgt = np.einsum("rk,rkc->rc", w, tv[rng.integers(0, T, R)]).astype(np.float32)
        o = (tgt + rng.normal(size=(R, 3)).astype(np.float32) * 5 * scale).astype(np.float32)
        d = ((tgt - o) * np.float32(rng.choice([1.0, 2.0, 0.5]))).astype(np.float32)
    elif kind == 3:  # degenerate triangles + rays lying in z = 0
        tv = (rng.integers(-3, 4, (T, 3, 3))).astype(np.float32)
        tv[: T // 3, :, 2] = 0
        o = rng.integers(-3, 4, (R, 3)).astype(np.float32)
        d = rng.integers(-3, 4, (R, 3)).astype(np.float32)
        o[: R // 2, 2] = 0
        d[: R // 2, 2] = 0
    else:  # extreme magnitudes (reciprocal slow path: denormal / huge determinants)
        ex = rng.uniform(-20, 18, (T, 1, 1))
        tv = (rng.normal(size=(T, 3, 3)) * 10.0 ** ex).astype(np.float32)
        o = (rng.normal(size=(R, 3)) * 10.0 ** rng.uniform(-20, 18, (R, 1))).astype(np.float32)
        d = (rng.normal(size=(R, 3)) * 10.0 ** rng.uniform(-20, 18, (R, 1))).astype(np.float32)
    u = rng.random()
    eps = None if u < 0.65 else (float(10.0 ** rng.uniform(-8, -1)) if u < 0.93 else float(rng.choice([0.0, 1e-42, 1e-38])))
    with np.errstate(all="ignore"):
        et, eh = orc.ray_intersect_triangle_dense(o, d, tv, epsilon=eps)
    t, hit = G.ray_intersect_triangle(torch.as_tensor(o, device="cuda")[:, None, :], torch.as_tensor(d, device="cuda")[:, None, :],
                                      torch.as_tensor(tv, device="cuda"), epsilon=eps)
    gt, gh = t.cpu().numpy(), hit.cpu().numpy()
    st["cases"] += 1
    st["tests"] += R * tv.shape[0]
    st["hits"] += int(eh.sum())
    st["hit_mismatch"] += int((gh != eh).sum())
    same = (gt.view(np.uint32) == et.view(np.uint32)) | (np.isnan(gt) & np.isnan(et))
    st["t_mismatch"] += int((~same).sum())
st["seconds"] = time.time() - t0
print(json.dumps(st))
