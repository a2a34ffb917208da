"""Replay scenes saved by scratch/beam_stress.py: a lost path (missed_case*.npz: which variants lose it) or two mappings
whose candidate rows differ (case*_<mapping>.npz: the row counts of every mapping, kappa 64 and 4096).
python scratch/beam_replay_r06.py scratch/cases_r06/*.npz"""
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
    M = float(max(np.abs(V).max(), np.abs(tx).max(), np.abs(rx).max()))
    out = {"file": f.split("/")[-1], "order": order, "T": int(Tr.shape[0]), "ntx": len(tx), "nrx": len(rx), "valid": len(ea),
           "M": M, "ulp_M": float(np.spacing(np.float32(M)))}
    for name, kw in (("default", {}), ("k4096", {"kappa": 4096.0}), ("plain", {"expansion": "plain"}), ("fused", {"expansion": "fused"}),
                     ("emit_plain", {"emit": "plain"}), ("emit_clustered", {"emit": "clustered"}),
                     ("plain_emit_plain", {"expansion": "plain", "emit": "plain"})):
        if name == "fused" and order < 2:
            continue
        try:
            bp = tr.trace_beam_pruned(scene, order, **kw)
            ba = set(map(tuple, bp.objects.cpu().tolist()))
            out[name] = {"missed": sorted(ea - ba), "extra": len(ba - ea), "levels": tr.last_beam_stats["levels"],
                         "rows": tr.last_beam_stats["rows"]}
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": repr(exc)[:200]}
    print(json.dumps(out), flush=True)
