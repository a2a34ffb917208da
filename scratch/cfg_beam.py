"""Conservative beam pruning on the BASELINE configurations: configs[2] (order 2), configs[3] (order 3, FULL
coverage of the 1.02e15 candidates) and configs[4] (200k triangles, 1024 RX, order 2), forward + grad(TX).
python scratch/cfg_beam.py [cfg3] [cfg4] [cfg5] [--expansion=auto|clustered|plain] [--kappa=64] [--emit=auto|plain|clustered]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

which = [a for a in sys.argv[1:] if a.startswith(("cfg", "bruxelles"))] or ["cfg3", "cfg4"]
EXPANSION = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--expansion=")), "auto")
EMIT = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--emit=")), "auto")
KAPPA = float(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--kappa=")), "64"))
QUADS = "--quads" in sys.argv[1:]  # assume_quads=True, as the reference harness (tests/benchmarks/test_rt.py:151-196)


def run(name, V, Tr, tx, rx, order, verify=None):
    mesh = G.Mesh(V, Tr, assume_quads=QUADS)
    tracer = G.ExhaustivePathTracer(accel="bvh")  # occlusion stage on the LBVH (bit-identical, O(log T))

    def step():
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
        p = tracer.trace_beam_pruned(scene, order, expansion=EXPANSION, emit=EMIT, kappa=KAPPA)
        if p.objects.shape[0]:
            torch.sqrt((torch.diff(p.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
        return p, txg.grad

    p, g = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p, g = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = mesh.num_primitives
    out = {"config": name, "order": order, "triangles": int(Tr.shape[0]), "num_tx": len(tx), "num_rx": len(rx),
           "exhaustive_candidates": len(tx) * len(rx) * n * (n - 1) ** (order - 1), "s_per_step": dt,
           "valid_paths": int(p.objects.shape[0]), "expansion": EXPANSION, "emit": EMIT, "kappa": KAPPA, "assume_quads": QUADS, **tracer.last_beam_stats,
           "grad_finite": bool(torch.isfinite(g).all()) if g is not None else None}
    if verify is not None:
        ex = verify(mesh)
        a = set(map(tuple, ex.objects.cpu().numpy().tolist()))
        b = set(map(tuple, p.objects.cpu().numpy().tolist()))
        out.update({"exhaustive_valid": len(a), "missed": len(a - b), "extra": len(b - a)})
    print(json.dumps(out), flush=True)


if "cfg3" in which or "cfg4" in which:
    V, Tr, c, h = S.manhattan(1000)
    tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
    if "cfg3" in which:
        def ver(mesh):
            scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
            return G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, max_survivors=1 << 24, max_paths=1 << 20)
        run("configs[2]", V, Tr, tx, rx, 2, ver)
    if "cfg4" in which:
        run("configs[3]", V, Tr, tx, rx, 3)
if "cfg5" in which:
    V, Tr, tx, rx = S.cfg5_scene()
    run("configs[4]", V, Tr, tx, rx, 2)
for w in which:  # the reference's own mesh (tests/golden/bruxelles.npz): 16 TX x 64 RX in the open, order 2 / 3
    if w.startswith("bruxelles"):
        V, Tr = S.load_real_mesh("bruxelles")
        tx, rx = S.outdoor_end_points(G, V, Tr, 16, 64)
        run(f"bruxelles order {w[-1]}", V, Tr, tx, rx, int(w[-1]))
