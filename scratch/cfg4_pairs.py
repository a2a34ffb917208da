"""BASELINE configs[3] (order 3, 10k-triangle scene, 16 TX x 64 RX): FULL coverage of the candidate space
pruned by per-pair visibility (HybridPathTracer.trace_pairs), forward + gradient w.r.t. TX.
Prints the size of the pruned space first and refuses to run more than `budget` candidate evaluations.
python scratch/cfg4_pairs.py [order] [budget_evals] [--samples] [--strategy=auto|ragged|prefix|loop]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

POS = [a for a in sys.argv[1:] if not a.startswith("--")]
order = int(POS[0]) if POS else 3
budget = float(POS[1]) if len(POS) > 1 else 5e12
V, Tr, centres, heights = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(centres, heights, 16, 64)
mesh = G.Mesh(V, Tr)
STRATEGY = next((a.split("=")[1] for a in sys.argv if a.startswith("--strategy=")), "auto")
SAMPLES = "--samples" in sys.argv
NUM_RAYS = int(float(next((a.split("=")[1] for a in sys.argv if a.startswith("--rays=")), 1e6)))
solver = G.HybridPathTracer(num_rays=NUM_RAYS, accel="bvh", pairs_strategy=STRATEGY, sample_triangles=SAMPLES)
vt = mesh.triangles_visible_from_vertex(torch.tensor(tx, device="cuda"), num_rays=NUM_RAYS, accel="bvh", sample_triangles=SAMPLES).sum(1)
vr = mesh.triangles_visible_from_vertex(torch.tensor(rx, device="cuda"), num_rays=NUM_RAYS, accel="bvh", sample_triangles=SAMPLES).sum(1)
evals = int((vt.double().sum() * vr.double().sum()).item()) * mesh.num_primitives ** (order - 2)
out = {"num_rays": NUM_RAYS, "sample_triangles": SAMPLES, "strategy": STRATEGY, "order": order, "visible_per_tx_mean": float(vt.float().mean()), "visible_per_rx_mean": float(vr.float().mean()),
       "candidate_evals_per_step": evals, "exhaustive_evals_per_step": 1024 * 10000 * 9999 ** (order - 1)}
print(json.dumps(out), flush=True)
if evals > budget:
    sys.exit("over budget")


def step():
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
    paths = solver.trace_pairs(scene, order, max_survivors=1 << 22)
    torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
    return paths, txg.grad


paths, grad = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
paths, grad = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out.update({"s_per_step": dt, "valid_paths": int(paths.objects.shape[0]), "candidate_evals_per_s": evals / dt,
            "grad_finite": bool(torch.isfinite(grad).all())})
print(json.dumps(out))
