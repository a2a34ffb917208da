"""HybridPathTracer: the visibility-pruned candidate space unranked on the GPU (trace_rank_range, product
mode of drt_candidates) vs tracing the table enumerated by the host DiGraph, over random scenes / masks /
quads / orders; and trace_pairs (per-pair pruning) must return a subset of the exhaustive valid paths with
identical vertices.  python scratch/hybrid_stress.py [seconds]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(21)
st = {"cases": 0, "valid_paths": 0, "object_mismatch_cases": 0, "vertex_mismatch_cases": 0, "pairs_not_subset": 0,
      "pairs_vertex_mismatch": 0, "pairs_found": 0, "pairs_exhaustive": 0, "window_mismatch_cases": 0}
t0 = time.time()
while time.time() - t0 < budget:
    boxes = int(rng.integers(1, 9))
    pitch = float(rng.uniform(20, 45))
    V, Tr, c, h = S.manhattan(boxes, pitch=pitch, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.5:
        ext = float(np.abs(V[:, :2]).max()) + 10
        gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
        Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
        V = np.concatenate((V, gv))
    ntx, nrx = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    tx, rx = S.manhattan_tx_rx(c, h, min(ntx, boxes), nrx, seed=int(rng.integers(1 << 30)), pitch=pitch)
    tx[:, 2] = rng.uniform(2, 40, len(tx))
    if rng.random() < 0.5:  # round 5: rotated cities (any yaw, tilt <= 10 degrees)
        V, tx, rx = S.rotate_points(S.random_rotation(rng), V, tx, rx)
        st["rotated"] = st.get("rotated", 0) + 1
    quads = bool(rng.random() < 0.4)
    mask = (rng.random(Tr.shape[0]) > 0.15) if rng.random() < 0.5 else None
    if mask is not None and quads:
        mask[1::2] = mask[0::2]
    order = int(rng.choice([0, 1, 2, 2, 3]))
    n = Tr.shape[0] // 2 if quads else Tr.shape[0]
    if order == 3 and n > 40:
        order = 2
    scene = G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask, assume_quads=quads))
    solver = G.HybridPathTracer(num_rays=int(rng.choice([2000, 50_000, 300_000])))
    cands, _ = solver.generate_path_candidates(scene, order)
    ref = solver.trace_path_candidates_compact(scene, cands)
    got = scene.trace_paths(order, solver=solver, compact=True)
    ro, go = ref.objects.cpu().numpy(), got.objects.cpu().numpy()
    st["cases"] += 1
    st["valid_paths"] += len(ro)
    same = ro.shape == go.shape and (ro == go).all()
    st["object_mismatch_cases"] += int(not same)
    if same:
        st["vertex_mismatch_cases"] += int((ref.vertices.cpu().numpy().view(np.uint32) != got.vertices.cpu().numpy().view(np.uint32)).any())
    total = solver.num_path_candidates(scene, order)
    cut = int(rng.integers(0, total + 1))
    a, b = solver.trace_rank_range(scene, order, 0, cut), solver.trace_rank_range(scene, order, cut, total)
    both = sorted(map(tuple, np.concatenate((a.objects.cpu().numpy(), b.objects.cpu().numpy())).tolist()))
    st["window_mismatch_cases"] += int(both != sorted(map(tuple, ro.tolist())))
    if order >= 1:
        ex = scene.trace_paths(order, compact=True)
        solver.pairs_strategy = str(rng.choice(["ragged", "prefix", "loop", "auto"]))
        pp = solver.trace_pairs(scene, order)
        eo = {tuple(r): i for i, r in enumerate(ex.objects.cpu().numpy().tolist())}
        po = pp.objects.cpu().numpy().tolist()
        st["pairs_exhaustive"] += len(eo)
        st["pairs_found"] += len(po)
        idx = [eo.get(tuple(r), -1) for r in po]
        st["pairs_not_subset"] += sum(i < 0 for i in idx)
        if po and all(i >= 0 for i in idx):
            ev = ex.vertices.cpu().numpy()[idx]
            st["pairs_vertex_mismatch"] += int((ev.view(np.uint32) != pp.vertices.cpu().numpy().view(np.uint32)).any())
st["seconds"] = time.time() - t0
print(json.dumps(st))
