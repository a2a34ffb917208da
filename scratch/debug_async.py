import ctypes as C, sys
sys.path.insert(0, ".")
import torch
sys.path.insert(0, "tests")
from test_hipgraph_gpu import _trace_setup
from differt_amd._tensors import ptr, stream
C_, lib, mesh, tx, rx, params, cands, order = _trace_setup()
L = lib.load()
cap_s, cap_p = 1 << 16, 256
nb = L.drt_trace_compact_workspace_size(cap_s, cap_p)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
keys = torch.empty(cap_p, dtype=torch.int64, device="cuda")
verts = torch.empty((cap_p, order + 2, 3), dtype=torch.float32, device="cuda")
objs = torch.empty((cap_p, order + 2), dtype=torch.int32, device="cuda")
counts = torch.zeros(4, dtype=torch.int64, device="cuda")
h = mesh.handle().h
def launch():
    lib.call("drt_trace_paths_compact_async", h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx), rx.shape[0],
             C.byref(cands), cap_s, cap_p, ptr(keys), ptr(verts), ptr(objs), ptr(counts), ptr(ws), nb, stream())
launch(); torch.cuda.synchronize(); print("eager", counts.tolist(), keys[:12].tolist())
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    launch()
torch.cuda.synchronize(); print("side", counts.tolist())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    launch()
torch.cuda.synchronize()
for rep in range(3):
    counts.fill_(13); keys.fill_(13)
    g.replay(); torch.cuda.synchronize()
    print("replay", rep, counts.tolist(), keys[:12].tolist(), ws[:16].view(torch.int64).tolist())
