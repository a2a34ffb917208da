import sys, torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
scene = G.Scene(torch.as_tensor(tx, device="cuda"), torch.as_tensor(rx, device="cuda"), G.Mesh(V, Tr))
tr = G.ExhaustivePathTracer(accel="bvh")
caps = dict(max_records=1 << 24, max_rows=1 << 23, max_survivors=1 << 21)
for _ in range(3):
    tr.trace_beam_pruned(scene, 2)
torch.cuda.synchronize()
out = None
for _ in range(3):
    out = tr.trace_beam_pruned_static(scene, 2, max_paths=4096, out=out)
torch.cuda.synchronize()
out = None
for _ in range(3):
    out = tr.trace_beam_pruned_static(scene, 2, max_paths=4096, out=out, **caps)
torch.cuda.synchronize()
for _ in range(2):
    tr.trace_beam_pruned(scene, 2)
torch.cuda.synchronize()
