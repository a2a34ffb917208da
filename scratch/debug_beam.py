import sys
import numpy as np, torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S
rng = np.random.default_rng(1234)
order, assume_quads, masked = 1, False, True
for trial in range(3):
    boxes = int(rng.integers(4, 28))
    V, Tr, c, h = S.manhattan(boxes, pitch=float(rng.uniform(20, 45)), seed=int(rng.integers(1 << 30)))
    ext = float(np.abs(V[:, :2]).max()) + 10
    gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
    Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
    V = np.concatenate((V, gv))
    tx, rx = S.manhattan_tx_rx(c, h, 2, 5, seed=int(rng.integers(1 << 30)))
    tx[:, 2] = rng.uniform(2, 60, len(tx))
    mask = rng.random(Tr.shape[0]) > 0.15
    mesh = G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    tr = G.ExhaustivePathTracer()
    ex = tr.trace_rank_range(scene, order)
    bp = tr.trace_beam_pruned(scene, order)
    print(trial, boxes, Tr.shape[0], "exhaustive", ex.objects.tolist(), "beam", bp.objects.tolist(), tr.last_beam_stats)
    if ex.objects.shape[0] != bp.objects.shape[0]:
        # which test kills them?  re-run with a huge margin
        wide = tr.trace_beam_pruned(scene, order, kappa=1e9)
        print("  wide margin:", wide.objects.tolist(), tr.last_beam_stats)
        for o in ex.objects.tolist():
            t, a, r = o
            tv = mesh.triangle_vertices[a].cpu().numpy(); n = mesh.normals[a].cpu().numpy()
            print("  path", o, "tri", tv.tolist(), "n", n.tolist(), "tx", tx[t].tolist(), "rx", rx[r].tolist(),
                  "d_tx", float(np.dot(tx[t] - tv[0], n)), "d_rx", float(np.dot(rx[r] - tv[0], n)))
