"""Stress comparison of the tracer's two occlusion stages (brute-force LDS tiles = normative, LBVH =
opt-in) and of rank-window tiling: identical valid-path keys / vertices expected.

    python scratch/trace_stress.py [seconds]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(7)
st = {"cases": 0, "candidate_evals": 0, "valid_paths": 0, "key_mismatch_cases": 0, "vertex_mismatch_cases": 0}
t0 = time.time()
while time.time() - t0 < budget:
    boxes = int(rng.integers(20, 400))
    V, Tr, c, h = S.manhattan(boxes, pitch=float(rng.uniform(25, 50)), seed=int(rng.integers(1 << 30)))
    ntx, nrx = int(rng.integers(1, 6)), int(rng.integers(1, 12))
    tx, rx = S.manhattan_tx_rx(c, h, ntx, nrx, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.5:  # low TX -> more multi-bounce street-level paths
        tx[:, 2] = rng.uniform(2, 15, ntx)
    if rng.random() < 0.5:  # round 5: rotated cities (any yaw, tilt <= 10 degrees)
        V, tx, rx = S.rotate_points(S.random_rotation(rng), V, tx, rx)
        st["rotated"] = st.get("rotated", 0) + 1
    quads = bool(rng.random() < 0.3)
    mask = (rng.random(Tr.shape[0]) > 0.05) if rng.random() < 0.4 else None
    if mask is not None and quads:
        mask[1::2] = mask[0::2]
    mesh = G.Mesh(V, Tr, mask=mask, assume_quads=quads)
    scene = G.Scene(tx, rx, mesh)
    order = int(rng.choice([1, 2, 2, 3]))
    n = mesh.num_primitives
    total = n * (n - 1) ** (order - 1)
    cnt = int(min(total, rng.integers(1, 40) * 1_000_000))
    lo = int(rng.integers(0, total - cnt + 1))
    a = G.ExhaustivePathTracer().trace_rank_range_literal(scene, order, lo, lo + cnt, max_survivors=1 << 23)
    b = G.ExhaustivePathTracer(accel="bvh").trace_rank_range_literal(scene, order, lo, lo + cnt, max_survivors=1 << 23)
    st["cases"] += 1
    st["candidate_evals"] += cnt * ntx * nrx
    st["valid_paths"] += int(a.keys.shape[0])
    if a.keys.shape != b.keys.shape or not torch.equal(a.keys, b.keys):
        st["key_mismatch_cases"] += 1
    elif not torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32)):
        st["vertex_mismatch_cases"] += 1
st["seconds"] = time.time() - t0
print(json.dumps(st))
