"""Order-3 completeness of the per-pair pruned search on scenes small enough for the exhaustive tracer:
N-box Manhattan meshes, 4 TX x 16 RX.  python scratch/order3_completeness.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

rows = []
for boxes, seed in ((30, 1), (60, 2), (100, 3), (150, 4)):
    V, Tr, c, h = S.manhattan(boxes, seed=seed)
    tx, rx = S.manhattan_tx_rx(c, h, 4, 16, seed=seed + 10)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
    ex = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 3, max_survivors=1 << 24, max_paths=1 << 18)
    a = set(map(tuple, ex.objects.cpu().numpy().tolist()))
    row = {"triangles": int(Tr.shape[0]), "exhaustive_evals": 64 * Tr.shape[0] * (Tr.shape[0] - 1) ** 2, "exhaustive_valid": len(a)}
    for name, kw in (("lattice_1e6", {"num_rays": 1_000_000}), ("lattice_1e5_plus_samples", {"num_rays": 100_000, "sample_triangles": True})):
        solver = G.HybridPathTracer(accel="bvh", **kw)
        p = solver.trace_pairs(scene, 3, max_survivors=1 << 22, max_paths=1 << 18)
        b = set(map(tuple, p.objects.cpu().numpy().tolist()))
        row[name] = {"evals": int(solver.last_num_evaluated), "found": len(b), "missed": len(a - b), "extra": len(b - a)}
    rows.append(row)
    print(json.dumps(row), flush=True)
