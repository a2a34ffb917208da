"""Reproduce a rows mismatch between the clustered expansion (with the receiver-box child filter) and the plain one:
python scratch/filter_case.py tests/golden/beam_cases/filter_case99516.npz"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402

d = np.load(sys.argv[1])
mask = d["mask"] if d["mask"].size else None
mesh = G.Mesh(d["V"], d["Tr"], mask=mask, assume_quads=bool(d["assume_quads"]))
scene = G.Scene(torch.tensor(d["tx"], device="cuda"), torch.tensor(d["rx"], device="cuda"), mesh)
tr = G.ExhaustivePathTracer()
out = {}
for name, kw in (("auto", {}), ("plain", {"expansion": "plain"})):
    r = tr.trace_beam_pruned(scene, int(d["order"]), **kw)
    out[name] = {"rows": tr.last_beam_stats["rows"], "levels": tr.last_beam_stats["levels"], "valid": int(r.objects.shape[0])}
print(json.dumps(out))
