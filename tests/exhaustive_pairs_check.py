"""Helper of tests/test_full_size_gpu.py and scratch/exhaustive_pairs.py (not a test module): completeness of the pruned
("beam") search at the BASELINE configs' OWN size against the exhaustive tracer -- for selected (tx, rx) pairs,
`trace_rank_range` evaluates ALL candidates of the pair (reference: full enumeration, geometry/_solvers.py:803-848 +
_trace_path_candidates :499-770) in rank windows; objects / vertex bits must equal the rows of the full pruned result that
belong to the pair.  Pair selection (deterministic): the pairs with the most paths in the pruned result, pairs with none,
and the "grazing-heavy" pairs -- end points with the most triangle planes passing within 1/180 of their distance."""

from __future__ import annotations

import json
import time

import numpy as np
import torch

import differt_amd.geometry as G


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
        p = tracer.trace_rank_range_literal(scene, order, lo, min(lo + step, total), max_survivors=max_survivors, max_paths=1 << 16)
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
