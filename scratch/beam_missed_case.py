"""Replay scenes in which the pruned search lost a path (saved by scratch/beam_stress.py): which variants lose it.
python scratch/beam_missed_case.py gpurun_out/stress_mismatch/missed_case*.npz"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402

for f in sys.argv[1:]:
    d = np.load(f)
    V, Tr, tx, rx, order = d["V"], d["Tr"], d["tx"], d["rx"], int(d["order"])
    mask = d["mask"] if d["mask"].size else None
    quads = bool(d["assume_quads"])
    mesh = G.Mesh(V, Tr, mask=mask, assume_quads=quads)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    tr = G.ExhaustivePathTracer()
    ex = tr.trace_rank_range_literal(scene, order, max_survivors=1 << 24, max_paths=1 << 20)
    ea = set(map(tuple, ex.objects.cpu().tolist()))
    out = {"file": f.split("/")[-1], "order": order, "T": int(Tr.shape[0]), "quads": quads, "masked": mask is not None, "ntx": len(tx),
           "nrx": len(rx), "valid": len(ea), "saved_missed": d["missed"].tolist(), "axis_aligned": bool(np.all(np.isin(np.abs(mesh.handle().normals().cpu().numpy()), (0.0, 1.0))))}
    for name, kw in (("default", {}), ("k256", {"kappa": 256.0}), ("k4096", {"kappa": 4096.0}), ("k1e6", {"kappa": 1e6}), ("nopairs", {"pairs": False}),
                     ("plain", {"expansion": "plain"}), ("plain_nopairs", {"expansion": "plain", "pairs": False}),
                     ("emit_plain", {"emit": "plain"}), ("emit_clustered", {"emit": "clustered"}), ("rows_plain", {"rows": "plain"})):
        try:
            bp = tr.trace_beam_pruned(scene, order, **kw)
            ba = set(map(tuple, bp.objects.cpu().tolist()))
            out[name] = {"missed": sorted(ea - ba), "extra": len(ba - ea), "pair_mode": tr.last_beam_stats["pair_mode"],
                         "levels": tr.last_beam_stats["levels"], "rows": tr.last_beam_stats["rows"]}
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": repr(exc)[:200]}
    # geometry of the lost path: incidence cosines at its mirrors, distances of its end points from the mirror planes
    for ob in d["missed"].tolist()[:2]:
        it, ir = ob[0], ob[-1]
        tri = ob[1:-1]
        m = (ex.objects[:, 0] == it) & (ex.objects[:, -1] == ir)
        for j, t in enumerate(tri):
            m = m & (ex.objects[:, j + 1] == t)
        pv = ex.vertices[m][0].double().cpu().numpy()
        nr = mesh.handle().normals().cpu().numpy().astype(np.float64)
        cosines = []
        for j, t in enumerate(tri):
            a, b = pv[j] - pv[j + 1], pv[j + 2] - pv[j + 1]
            cosines.append([float(abs(a @ nr[t]) / np.linalg.norm(a)), float(abs(b @ nr[t]) / np.linalg.norm(b))])
        out.setdefault("lost_paths", []).append({"objects": ob, "vertices": pv.tolist(), "cos_incidence": cosines,
                                                 "segment_lengths": np.linalg.norm(np.diff(pv, axis=0), axis=1).tolist(),
                                                 "tri_verts": V[Tr[tri]].tolist()})
    print(json.dumps(out), flush=True)
