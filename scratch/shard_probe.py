import sys, time, json, torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S
V, Tr, tx, rx = S.cfg5_scene()
mesh = G.Mesh(V, Tr)
tr = G.ExhaustivePathTracer(accel="bvh")
def step(shard):
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    sc = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
    p = tr.trace_beam_pruned(sc, 2, prefix_shard=shard)
    if p.objects.shape[0]:
        torch.sqrt((torch.diff(p.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
    return p
for shard in (None, (0, 8), (3, 8)):
    step(shard); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); step(shard); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = tr.last_beam_stats
    print(shard, "wall ms", [round(t * 1e3, 2) for t in ts], "kernels", round(st["expand_last_ms"], 2), round(st["emit_ms"], 2), round(st["trace_ms"], 2), "slices", st.get("chunks"), st["levels"])
