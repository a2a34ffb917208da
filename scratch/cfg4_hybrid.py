"""BASELINE configs[3] (order 3 on the 10k-triangle scene, 16 TX x 64 RX) with FULL coverage of the
visibility-pruned candidate space (HybridPathTracer.trace_rank_range: F x N x L unranked on the GPU),
forward + gradient w.r.t. TX.  python scratch/cfg4_hybrid.py [order] [num_rays]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
num_rays = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
V, Tr, centres, heights = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(centres, heights, 16, 64)
mesh = G.Mesh(V, Tr)
solver = G.HybridPathTracer(num_rays=num_rays, accel="bvh")
out = {"order": order, "num_rays": num_rays}


def step():
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
    paths = solver.trace_rank_range(scene, order, max_survivors=1 << 25, max_paths=1 << 20)
    loss = torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum()
    loss.backward()
    return paths, txg.grad, scene


paths, grad, scene = step()
torch.cuda.synchronize()
first, last, middle, both = solver._visible_sets(scene)
out["visible_first"], out["visible_last"] = int(first.shape[0]), int(last.shape[0])
out["pruned_candidates_per_pair"] = solver.num_path_candidates(scene, order)
out["exhaustive_candidates_per_pair"] = 10000 * 9999 ** (order - 1)
t0 = time.perf_counter()
paths, grad, _ = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out["s_per_step"] = dt
out["valid_paths"] = int(paths.objects.shape[0])
out["candidate_evals_per_step"] = out["pruned_candidates_per_pair"] * 1024
out["candidate_evals_per_s"] = out["candidate_evals_per_step"] / dt
out["grad_finite"] = bool(torch.isfinite(grad).all())
print(json.dumps(out))
