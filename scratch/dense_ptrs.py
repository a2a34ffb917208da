"""kernel time of the dense tracer per chunk next to the addresses of its outputs (caching allocator generations)"""
import sys
import torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
mesh = G.Mesh(V, Tr)
scene = G.Scene(torch.tensor(tx[:1], device="cuda"), torch.tensor(rx, device="cuda"), mesh)
solver = G.ExhaustivePathTracer(chunk_size=1 << 20, collect_stats=True)
keep = []
for i, p in enumerate(scene.trace_paths(order=2, solver=solver)):
    if i >= 12:
        break
    ms = solver.last_stats["filter_ms"]
    ptrs = [t.data_ptr() for t in (p.vertices, p.objects, p.mask, p.interaction_types)]
    print(i, f"{ms:.3f}", [hex(x) for x in ptrs], [hex(x % (1 << 30)) for x in ptrs])
    if i % 3 == 2:
        keep.append(p)  # perturb the allocator: keep some generations alive
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
