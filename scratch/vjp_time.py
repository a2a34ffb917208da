"""Throughput of the differentiable hard-mode t (dense ray_intersect_triangle fwd + VJP) on the cfg2 geometry."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import differt_amd.geometry as G
from bench import make_cfg2
for R in (256, 4096):
    T = 10000
    o, d, tv = (torch.as_tensor(x, device="cuda") for x in make_cfg2(R, T, seed=5))
    o.requires_grad_(True); d.requires_grad_(True); tv.requires_grad_(True)
    def step():
        t, hit = G.ray_intersect_triangle(o[:, None, :], d[:, None, :], tv)
        loss = torch.where(hit, t, torch.zeros_like(t)).sum()
        g = torch.autograd.grad(loss, (o, d, tv))
        return g
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    def fwd():
        with torch.no_grad(): G.ray_intersect_triangle(o[:, None, :], d[:, None, :], tv)
    fwd(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fwd()
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 5
    print({"R": R, "T": T, "fwd_ms": df * 1e3, "fwd_bwd_ms": dt * 1e3, "tests_per_s_fwd_bwd": R * T / dt})
