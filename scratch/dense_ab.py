"""A/B of the two instantiations of mt_dense_aligned_kernel on the bench shape (same box, same buffers):
eps = reference default (EPS_COVERS) vs eps = 0 (general).  Prints min/median ms of 200 launches each."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import differt_amd.geometry as G
R, T = 65536, 10000
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.rand(R, 1, 3, device="cuda", generator=g) * 100 - 50
d = torch.rand(R, 1, 3, device="cuda", generator=g) * 100 - 50 - o
c = torch.rand(T, 1, 3, device="cuda", generator=g) * 100 - 50
tv = c + torch.cat([torch.zeros(T, 1, 3, device="cuda"), torch.randn(T, 2, 3, device="cuda", generator=g) * 2], 1)
from differt_amd import _lib
from differt_amd._tensors import stream
t = torch.empty(R, T, device="cuda"); h = torch.empty(R, T, dtype=torch.uint8, device="cuda")
oo = o.reshape(R, 3).contiguous(); dd = d.reshape(R, 3).contiguous()
def run(eps, n):
    for _ in range(n):
        _lib.call("drt_ray_intersect_triangle_dense", oo.data_ptr(), dd.data_ptr(), R, tv.data_ptr(), T, eps, t.data_ptr(), h.data_ptr(), stream())
for rnd in range(3):
    for eps in (1.1920929e-6, 0.0):
        run(eps, 30); torch.cuda.synchronize()
        ms = []
        for _ in range(10):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); run(eps, 20); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1) / 20)
        print(f"eps={eps:g} min {min(ms):.4f} med {sorted(ms)[5]:.4f} ms", flush=True)
