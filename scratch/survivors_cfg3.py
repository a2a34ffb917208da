import sys; sys.path.insert(0,'/root/repo')
import ctypes as C, numpy as np, torch
import differt_amd.geometry as G, synthetic_scenes as S
from differt_amd import _lib
from differt_amd._tensors import ptr, stream
from differt_amd.geometry._solvers import _params, _rank_candidates
V,Tr,c,h=S.manhattan(1000); tx,rx=S.manhattan_tx_rx(c,h,16,64)
mesh=G.Mesh(V,Tr); n=10000; total=n*(n-1)
txg=torch.tensor(tx,device='cuda'); rxg=torch.tensor(rx,device='cuda')
lib=_lib.load()
nb=lib.drt_trace_compact_workspace_size(16,16)
ws=torch.empty(nb,dtype=torch.uint8,device='cuda'); keys=torch.empty(16,dtype=torch.int64,device='cuda')
v=torch.empty((16,4,3),device='cuda'); o=torch.empty((16,4),dtype=torch.int32,device='cuda'); nv=C.c_int64(0)
cands=_rank_candidates(2,0,total,n,None); pr=_params(None,None,None)
rc=lib.drt_trace_paths_compact(mesh.handle().h,C.byref(pr),ptr(txg),16,ptr(rxg),64,C.byref(cands),16,16,ptr(keys),ptr(v),ptr(o),C.byref(nv),ptr(ws),nb,stream())
print("rc",rc,"survivors of the geometric checks:",nv.value, lib.drt_last_error())
