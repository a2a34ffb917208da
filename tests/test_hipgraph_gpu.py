"""The asynchronous entry points are HIP-graph capturable (no hidden sync / allocation): capture a
sequence of launches on torch's capture stream, replay it, compare with eager results."""

from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_capture_and_replay_ray_queries():
    import differt_amd._lib as lib
    from differt_amd._tensors import ptr, stream

    rng = np.random.default_rng(0)
    R, T = 512, 3000
    o = torch.as_tensor(rng.normal(size=(R, 3)).astype(np.float32) * 5, device="cuda")
    tv = torch.as_tensor(rng.normal(size=(T, 3, 3)).astype(np.float32) * 2, device="cuda")
    d = tv.mean(dim=1)[torch.randint(0, T, (R,), device="cuda")] - o
    eps, tol = 10 * 1.1920929e-7, 100 * 1.1920929e-7
    t = torch.empty((R, T), device="cuda")
    hit = torch.empty((R, T), dtype=torch.uint8, device="cuda")
    blocked = torch.empty(R, dtype=torch.uint8, device="cuda")
    idx = torch.empty(R, dtype=torch.int32, device="cuda")
    tmin = torch.empty(R, device="cuda")
    ws = torch.empty(R, dtype=torch.int64, device="cuda")

    def launch():
        lib.call("drt_ray_intersect_triangle_dense", ptr(o), ptr(d), R, ptr(tv), T, eps, ptr(t), ptr(hit), stream())
        lib.call("drt_ray_intersect_any_triangle", ptr(o), ptr(d), R, ptr(tv), T, 0, None, 0, eps, tol,
                 ptr(blocked), stream())
        lib.call("drt_first_triangle_hit_by_ray", ptr(o), ptr(d), R, ptr(tv), T, 0, None, 0, eps, 512,
                 ptr(idx), ptr(tmin), ptr(ws), R * 8, stream())

    launch()
    torch.cuda.synchronize()
    ref = [x.clone() for x in (t, hit, blocked, idx, tmin)]
    for x in (t, hit, blocked, idx, tmin):
        x.zero_()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch()  # warm-up on the side stream
    torch.cuda.synchronize()
    for x in (t, hit, blocked, idx, tmin):
        x.zero_()
    with torch.cuda.graph(g):
        launch()
    for x in (t, hit, blocked, idx, tmin):
        x.zero_()
    g.replay()
    torch.cuda.synchronize()
    for got, exp in zip((t, hit, blocked, idx, tmin), ref):
        assert torch.equal(got, exp)
    assert int(ref[2].sum()) > 0
