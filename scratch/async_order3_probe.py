"""Validation: the one-pass entry point (drt_trace_paths_beam_async) on configs[3] at full size -- capacities of twice what
the synchronous call measures -- must return the synchronous call's paths.  python scratch/async_order3_probe.py"""
import json, sys, time, torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
mesh = G.Mesh(V, Tr)
tr = G.ExhaustivePathTracer(accel="bvh")
sc = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
ref = tr.trace_beam_pruned(sc, 3)
st = tr.last_beam_stats
p2 = lambda v: 1 << max(int(v) - 1, 1).bit_length()
caps = {"max_entries": p2(2 * st["levels"][1]), "max_records": p2(2 * st["levels"][2]), "max_rows": p2(2 * st["rows"]),
        "max_survivors": p2(st["rows"] // 2)}
out = tr.trace_beam_pruned_static(sc, 3, max_paths=4096, **caps)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = tr.trace_beam_pruned_static(sc, 3, max_paths=4096, out=out, **caps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = ref.objects.shape[0]
cts = out["counts"].tolist()
print(json.dumps({"config": "configs[3] through drt_trace_paths_beam_async (one pass)", "capacities": caps, "counts": cts,
                  "seconds": dt, "workspace_GiB": out["workspace"].numel() / 2 ** 30, "sync_valid_paths": n,
                  "same_keys": bool(cts[1] == n and torch.equal(out["keys"][:n], ref.keys)),
                  "same_vertex_bits": bool(cts[1] == n and torch.equal(out["vertices"][:n].view(torch.int32), ref.vertices.view(torch.int32)))}))
