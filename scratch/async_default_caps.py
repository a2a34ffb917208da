import sys, time, torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
mesh = G.Mesh(V, Tr)
tr = G.ExhaustivePathTracer(accel="bvh")
sc = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
for caps in ({}, {"max_records": 1 << 24, "max_rows": 1 << 23, "max_survivors": 1 << 21}):
    out = tr.trace_beam_pruned_static(sc, 2, max_paths=4096, **caps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.trace_beam_pruned_static(sc, 2, max_paths=4096, out=out, **caps)
    torch.cuda.synchronize()
    print(caps, "ms per call", (time.perf_counter() - t0) / 5 * 1e3, out["counts"].tolist())
