"""Fused tracer (dense + compact) vs the CPU oracle over random scenes: masks, objects and vertices
bit for bit; half of the cities rotated (round 5), half of the scenes triangle soups (round 6).
    python scratch/trace_oracle_stress.py [seconds] [--no-rotate] [--no-soup]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import oracle as orc  # noqa: E402
import synthetic_scenes as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 60.0
rng = np.random.default_rng(31)
st = {"cases": 0, "candidate_evals": 0, "valid_paths": 0, "mask_mismatch": 0, "vertex_mismatch": 0,
      "object_mismatch": 0, "compact_mismatch": 0}
t0 = time.time()
while time.time() - t0 < budget:
    ntx, nrx = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    # round 6: half of the scenes are triangle SOUPS (synthetic_scenes.soup_city: ear-clipped / gable / hip roofs, slivers,
    # T-junctions, duplicated vertices, a uniform 3-D rotation, 1e4-1e5 m offsets)
    soup = "--no-soup" not in sys.argv and rng.random() < 0.5
    if soup:
        V, Tr, info = S.soup_city(rng, int(rng.integers(1, 12)), extent=float(rng.uniform(40, 200)))
        tx, rx = S.soup_end_points(rng, V, ntx, nrx)
        st["soups"] = st.get("soups", 0) + 1
        st["soup_far"] = st.get("soup_far", 0) + int(info["offset_m"] > 0)
    else:
        boxes = int(rng.integers(2, 40))
        pitch = float(rng.uniform(20, 45))
        V, Tr, c, h = S.manhattan(boxes, pitch=pitch, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.5:  # add a ground quad under the city
            ext = float(np.abs(V[:, :2]).max()) + 10
            gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
            Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
            V = np.concatenate((V, gv))
        tx, rx = S.manhattan_tx_rx(c, h, min(ntx, boxes), nrx, seed=int(rng.integers(1 << 30)), pitch=pitch)
        tx[:, 2] = rng.uniform(2, 40, len(tx))
        if "--no-rotate" not in sys.argv and rng.random() < 0.5:  # round 5: rotated cities (any yaw, tilt <= 10 degrees)
            V, tx, rx = S.rotate_points(S.random_rotation(rng), V, tx, rx)
            st["rotated"] = st.get("rotated", 0) + 1
    quads = bool(rng.random() < 0.3) and Tr.shape[0] % 2 == 0  # (assume_quads pairs consecutive triangles whatever they are)
    mask = (rng.random(Tr.shape[0]) > 0.1) if rng.random() < 0.4 else None
    if mask is not None and quads:
        mask[1::2] = mask[0::2]
    order = int(rng.choice([0, 1, 2, 2, 3]))
    n = Tr.shape[0] // 2 if quads else Tr.shape[0]
    full = orc.generate_all_path_candidates(n, order)
    if full.shape[0] > 20000:
        full = full[np.sort(rng.choice(full.shape[0], 20000, replace=False))]
    if soup and order >= 1 and full.shape[0]:
        # a random sample of a soup's candidates is all misses: half of the rows come from the valid paths (and their
        # neighbours in the table) that the GPU's own exhaustive tracer finds -- the oracle then decides every one of them
        n_tri = Tr.shape[0]
        if not quads and mask is None and n_tri * max(n_tri - 1, 1) ** (order - 1) < 4e9:
            ex = G.ExhaustivePathTracer().trace_rank_range_literal(G.Scene(tx, rx, G.Mesh(V, Tr)), order, max_survivors=1 << 22, max_paths=1 << 16)
            near = ex.objects.cpu().numpy()[:, 1:-1]
            if len(near):
                jig = near[rng.integers(0, len(near), 2000)].copy()
                sel = rng.random(jig.shape) < 0.2
                jig[sel] = rng.integers(0, n_tri, int(sel.sum()))
                ok = np.ones(len(jig), bool)
                for q in range(order - 1):  # (rows of the complete graph never name one triangle twice in a row)
                    ok &= jig[:, q] != jig[:, q + 1]
                jig = jig[ok]
                full = np.unique(np.concatenate((full, near, jig)), axis=0)
                st["soup_rows_near_valid_paths"] = st.get("soup_rows_near_valid_paths", 0) + len(near)
    cand = (full * (2 if quads else 1)).astype(np.int32)
    if rng.random() < 0.1 and order:
        cand[rng.integers(0, len(cand))] = -1  # a padding row
    ocand = cand - cand % 2 if quads else cand  # Scene.trace_paths rounds user-supplied ids (SC:756-757)
    o = orc.trace_path_candidates(V, Tr, tx, rx, ocand, mask=mask, assume_quads=quads)
    scene = G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask, assume_quads=quads))
    got = scene.trace_paths(path_candidates=cand)
    cp = scene.trace_paths(path_candidates=cand, compact=True)
    m = got.mask.cpu().numpy()
    st["cases"] += 1
    st["candidate_evals"] += m.size
    st["valid_paths"] += int(o["mask"].sum())
    st["mask_mismatch"] += int((m != o["mask"]).sum())
    st["vertex_mismatch"] += int((got.vertices.cpu().numpy().view(np.uint32) != o["vertices"].view(np.uint32)).sum())
    st["object_mismatch"] += int((got.objects.cpu().numpy() != o["objects"]).sum())
    st["compact_mismatch"] += int(not np.array_equal(cp.keys.cpu().numpy(), np.flatnonzero(o["mask"].reshape(-1))))
st["seconds"] = time.time() - t0
print(json.dumps(st))
