"""Completeness of the pruned ("beam") search at the BASELINE configs' OWN size, against the exhaustive tracer.

    python scratch/exhaustive_pairs.py [--pairs3 4] [--pairs5 4] [--out profiles/r04/stress/exhaustive_pairs.json]

configs[3] (order 3, 10 000 triangles): for each selected (tx, rx) pair, `trace_rank_range` evaluates ALL
n(n-1)^2 = 9.998e11 candidates of that pair (reference: full enumeration, geometry/_solvers.py:803-848 +
_trace_path_candidates :499-770) in rank windows; the objects / vertex bits it finds must equal the rows of the
full 16 x 64 beam result (kappa = 64 AND kappa = 1) that belong to the pair.  configs[4] (order 2, 200 000
triangles, 1 x 1024): the same with n(n-1) = 4.0e10 candidates per pair.

Pair selection (deterministic): the pairs with the most paths in the beam result, pairs with none, and the
"grazing-heavy" pairs -- end points with the most triangle planes passing within 1/180 of their distance (the
incidences that widen the search's margins, profiles/r03/beam.md).
"""

from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402


def grazing_score(points, V, Tr):
    """per point: number of triangles whose plane passes within (distance to the triangle) / 180 of it"""
    tv = V[Tr]  # [T,3,3]
    nrm = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 1])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    c = tv.mean(axis=1)
    out = []
    for p in points:
        h = np.abs(((p[None] - tv[:, 0]) * nrm).sum(1))
        d = np.linalg.norm(c - p[None], axis=1)
        out.append(int((h < d / 180.0).sum()))
    return np.asarray(out)


def select_pairs(bp_objects, ntx, nrx, tx, rx, V, Tr, count):
    o = bp_objects
    per = np.zeros((ntx, nrx), np.int64)
    np.add.at(per, (o[:, 0], o[:, -1]), 1)
    gt, gr = grazing_score(tx, V, Tr), grazing_score(rx, V, Tr)
    graz = gt[:, None] + gr[None, :]
    order_paths = np.dstack(np.unravel_index(np.argsort(-per, axis=None, kind="stable"), per.shape))[0]
    order_graz = np.dstack(np.unravel_index(np.argsort(-graz, axis=None, kind="stable"), graz.shape))[0]
    empties = np.argwhere(per == 0)
    picks, why = [], []
    srcs = [("most paths", order_paths), ("grazing-heavy end points", order_graz), ("no path in the pruned result", empties)]
    i = 0
    while len(picks) < count and any(len(s) > i for _, s in srcs):
        for name, s in srcs:
            if len(s) > i and len(picks) < count:
                p = (int(s[i][0]), int(s[i][1]))
                if p not in picks:
                    picks.append(p)
                    why.append(f"{name} ({int(per[p])} paths, grazing score {int(graz[p])})")
        i += 1
    return picks, why


def exhaustive_pair(mesh, tx1, rx1, order, n, windows, max_survivors):
    tracer = G.ExhaustivePathTracer()
    scene = G.Scene(torch.tensor(tx1[None], device="cuda"), torch.tensor(rx1[None], device="cuda"), mesh)
    total = n * (n - 1) ** (order - 1)
    step = -(-total // windows)
    objs, verts = [], []
    for lo in range(0, total, step):
        p = tracer.trace_rank_range(scene, order, lo, min(lo + step, total), max_survivors=max_survivors, max_paths=1 << 16)
        objs.append(p.objects.cpu().numpy())
        verts.append(p.vertices.cpu().numpy())
    return np.concatenate(objs), np.concatenate(verts), total


def check_config(name, V, Tr, tx, rx, order, npairs, kappas, windows, max_survivors):
    mesh = G.Mesh(V, Tr)
    n = mesh.num_primitives
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    tracer = G.ExhaustivePathTracer(accel="bvh")
    beams = {}
    for kappa in kappas:
        t0 = time.perf_counter()
        bp = tracer.trace_beam_pruned(scene, order, kappa=kappa)
        torch.cuda.synchronize()
        beams[kappa] = (bp.objects.cpu().numpy(), bp.vertices.cpu().numpy(), time.perf_counter() - t0)
    picks, why = select_pairs(beams[kappas[0]][0], tx.shape[0], rx.shape[0], tx, rx, V, Tr, npairs)
    rec = {"config": name, "order": order, "triangles": int(Tr.shape[0]), "pairs": [], "kappas": list(kappas),
           "beam_valid_paths": {str(k): int(v[0].shape[0]) for k, v in beams.items()},
           "beam_seconds": {str(k): v[2] for k, v in beams.items()}}
    ok = True
    for (it, ir), reason in zip(picks, why):
        t0 = time.perf_counter()
        eo, ev, total = exhaustive_pair(mesh, tx[it], rx[ir], order, n, windows, max_survivors)
        dt = time.perf_counter() - t0
        prec = {"tx": it, "rx": ir, "why": reason, "candidates": int(total), "exhaustive_seconds": dt,
                "exhaustive_valid_paths": int(eo.shape[0])}
        for kappa, (bo, bv, _) in beams.items():
            sel = (bo[:, 0] == it) & (bo[:, -1] == ir)
            same = (sel.sum() == eo.shape[0] and np.array_equal(bo[sel][:, 1:-1], eo[:, 1:-1])
                    and np.array_equal(bv[sel].view(np.uint32), ev.view(np.uint32)))
            prec[f"equal_kappa_{kappa:g}"] = bool(same)
            ok = ok and same
        rec["pairs"].append(prec)
        print(json.dumps(prec), flush=True)
    rec["all_equal"] = bool(ok)
    rec["checked_pairs"] = len(picks)
    rec["candidates_evaluated"] = int(sum(p["candidates"] for p in rec["pairs"]))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs3", type=int, default=4)
    ap.add_argument("--pairs5", type=int, default=4)
    ap.add_argument("--out", default="profiles/r04/stress/exhaustive_pairs.json")
    a = ap.parse_args()
    out = {"what": "beam-pruned search == exhaustive tracer on whole candidate spaces of single (tx, rx) pairs at the "
                   "BASELINE configs' own size: objects and vertex bits", "records": []}
    if a.pairs3:
        V, Tr, c, h = S.manhattan(1000)
        tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
        out["records"].append(check_config("configs[3]", V, Tr, tx, rx, 3, a.pairs3, (64.0, 1.0), 64, 1 << 24))
    if a.pairs5:
        V, Tr, tx, rx = S.cfg5_scene()
        out["records"].append(check_config("configs[4]", V, Tr, tx, rx, 2, a.pairs5, (64.0, 1.0), 4, 1 << 24))
    out["all_equal"] = all(r["all_equal"] for r in out["records"])
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({"all_equal": out["all_equal"], "out": a.out}))
    if not out["all_equal"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
