"""Level sizes / row counts / time of the beam pruning for several margins (small and full cities)."""
import json, sys, time
import torch
sys.path.insert(0, ".")
import differt_amd.geometry as G
import synthetic_scenes as S

boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
order = int(sys.argv[2]) if len(sys.argv) > 2 else 3
V, Tr, c, h = S.manhattan(boxes)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
mesh = G.Mesh(V, Tr)
scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
tr = G.ExhaustivePathTracer()
for kp in (4.0, 16.0, 64.0, 256.0):  # error unit u = kappa * ulp(M); rows / levels / time as a function of it
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = tr.trace_beam_pruned(scene, order, kappa=kp)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"boxes": boxes, "order": order, "kappa": kp, "s": dt, "valid": int(p.objects.shape[0]), **tr.last_beam_stats}), flush=True)
