"""BASELINE configs[4] geometry on ONE GPU: 1 TX, 32 x 32 RX grid, 200 000-triangle Manhattan mesh, order 2,
forward + gradient w.r.t. TX, FULL coverage of the per-pair visibility-pruned candidate space
(HybridPathTracer.trace_pairs).  python scratch/cfg5_pairs.py"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

V, Tr, centres, heights = S.manhattan(20000)
tx, _ = S.manhattan_tx_rx(centres, heights, 1, 1)
tx[0, :2] = centres.mean(axis=0)  # a tall mast in the middle of the city
tx[0, 2] = heights.max() + 10.0
c0 = centres.mean(axis=0)
g = (np.arange(32) - 15.5) * 40.0
rx = np.stack(np.meshgrid(c0[0] + g + 20.0, c0[1] + g + 20.0, indexing="ij"), -1).reshape(-1, 2)
rx = np.column_stack((rx, np.full(len(rx), 1.5))).astype(np.float32)
mesh = G.Mesh(V, Tr)
NUM_RAYS = int(float(next((a.split("=")[1] for a in sys.argv if a.startswith("--rays=")), 1e6)))
SAMPLES = "--samples" in sys.argv
solver = G.HybridPathTracer(num_rays=NUM_RAYS, accel="bvh", sample_triangles=SAMPLES)
out = {"sample_triangles": SAMPLES, "num_rays": NUM_RAYS, "triangles": int(Tr.shape[0]), "num_tx": 1, "num_rx": int(rx.shape[0]),
       "exhaustive_evals_per_step": int(rx.shape[0]) * 200000 * 199999}


def step():
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
    paths = solver.trace_pairs(scene, 2)
    torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
    return paths, txg.grad


paths, grad = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
paths, grad = step()
torch.cuda.synchronize()
out.update({"s_per_step": time.perf_counter() - t0, "valid_paths": int(paths.objects.shape[0]),
            "candidate_evals_per_step": int(solver.last_num_evaluated), "grad_finite": bool(torch.isfinite(grad).all()),
            "grad_absmax": float(grad.abs().max())})
print(json.dumps(out))
if "--verify" in sys.argv:
    # exhaustive trace of all 4.1e13 candidates (about two minutes): the pruned search must not miss a path
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    t0 = time.perf_counter()
    ex = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, max_survivors=1 << 24, max_paths=1 << 20)
    torch.cuda.synchronize()
    a = set(map(tuple, ex.objects.cpu().numpy().tolist()))
    b = set(map(tuple, paths.objects.cpu().numpy().tolist()))
    print(json.dumps({"exhaustive_s": time.perf_counter() - t0, "exhaustive_valid_paths": len(a),
                      "pruned_valid_paths": len(b), "missed_by_pruning": len(a - b), "extra_in_pruned": len(b - a)}))
