"""A/B of the asynchronous entry (static shapes) with the two-kernel and the one-kernel last expansion: configs[2] / [4].
python scratch/graph_ab.py"""
import json, sys, time
import torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S

def leg(name, V, Tr, tx, rx, order, caps):
    scene = G.Scene(torch.as_tensor(tx, device="cuda"), torch.as_tensor(rx, device="cuda"), G.Mesh(V, Tr))
    tr = G.ExhaustivePathTracer(accel="bvh")
    for exp in ("auto", "fused", "auto", "fused"):
        out = tr.trace_beam_pruned_static(scene, order, max_paths=4096, expansion=exp, **caps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = tr.trace_beam_pruned_static(scene, order, max_paths=4096, expansion=exp, out=out, **caps)
        torch.cuda.synchronize()
        print(json.dumps({"config": name, "expansion": exp, "ms": (time.perf_counter() - t0) * 100, "counts": out["counts"].tolist()}), flush=True)

V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
leg("configs[2]", V, Tr, tx, rx, 2, dict(max_records=1 << 24, max_rows=1 << 23, max_survivors=1 << 21))
V, Tr, c, h = S.manhattan(20000)
tx, rx = S.manhattan_tx_rx(c, h, 1, 1024)
leg("configs[4]", V, Tr, tx, rx, 2, dict(max_records=1 << 25, max_rows=1 << 27, max_survivors=1 << 25))
