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
    for rep in range(3):  # several replays: a graph must not depend on what the previous one left behind
        for x in (t, hit, blocked, idx, tmin):
            x.fill_(3)
        g.replay()
        torch.cuda.synchronize()
        for got, exp in zip((t, hit, blocked, idx, tmin), ref):
            assert torch.equal(got, exp), rep
    assert int(ref[2].sum()) > 0


def _trace_setup(num_boxes=60, ntx=3, nrx=5, order=2):
    import ctypes as C

    import differt_amd._lib as lib
    import differt_amd.geometry as G
    import synthetic_scenes as S
    from differt_amd.geometry._solvers import _params, _rank_candidates

    V, Tr, centres, heights = S.manhattan(num_boxes)
    tx, rx = S.manhattan_tx_rx(centres, heights, ntx, nrx)
    mesh = G.Mesh(V, Tr)
    n = mesh.num_primitives
    count = n * (n - 1) ** (order - 1)
    tx_d, rx_d = torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda")
    params = _params(None, None, None)
    cands = _rank_candidates(order, 0, count, n, None)
    return C, lib, mesh, tx_d, rx_d, params, cands, order


def test_trace_compact_async_matches_sync_and_flags_overflow():
    """drt_trace_paths_compact_async (static shapes, device-side counts, no host sync) returns the
    rows of drt_trace_paths_compact bit for bit, pads the rest, and flags both overflows."""
    from differt_amd._tensors import ptr, stream

    C, lib, mesh, tx, rx, params, cands, order = _trace_setup()
    L = lib.load()
    cap_s, cap_p = 1 << 16, 512

    def buffers(cp):
        return (torch.empty(cp, dtype=torch.int64, device="cuda"),
                torch.empty((cp, order + 2, 3), dtype=torch.float32, device="cuda"),
                torch.empty((cp, order + 2), dtype=torch.int32, device="cuda"))

    nb = L.drt_trace_compact_workspace_size(cap_s, cap_p)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    k0, v0, o0 = buffers(cap_p)
    nv = C.c_int64(0)
    lib.call("drt_trace_paths_compact", mesh.handle().h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx), rx.shape[0],
             C.byref(cands), cap_s, cap_p, ptr(k0), ptr(v0), ptr(o0), C.byref(nv), ptr(ws), nb, stream())
    n = int(nv.value)
    assert 0 < n < cap_p

    k1, v1, o1 = buffers(cap_p)
    for x in (k1, v1, o1):
        x.fill_(77)
    counts = torch.full((4,), -5, dtype=torch.int64, device="cuda")
    lib.call("drt_trace_paths_compact_async", mesh.handle().h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx),
             rx.shape[0], C.byref(cands), cap_s, cap_p, ptr(k1), ptr(v1), ptr(o1), ptr(counts), ptr(ws), nb, stream())
    torch.cuda.synchronize()
    c = counts.tolist()
    assert c[1] == n and c[2] == 0 and c[3] == 0 and c[0] >= n
    assert torch.equal(k1[:n], k0[:n]) and torch.equal(o1[:n], o0[:n])
    assert torch.equal(v1[:n].view(torch.int32), v0[:n].view(torch.int32))  # bit-identical vertices
    assert bool((k1[n:] == -1).all()) and bool((o1[n:] == -1).all()) and bool((v1[n:] == 0).all())

    # path-capacity overflow: status bit 2, count still exact, every written row is a valid path
    small = max(n // 2, 1)
    k2, v2, o2 = buffers(small)
    lib.call("drt_trace_paths_compact_async", mesh.handle().h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx),
             rx.shape[0], C.byref(cands), cap_s, small, ptr(k2), ptr(v2), ptr(o2), ptr(counts), ptr(ws), nb, stream())
    torch.cuda.synchronize()
    c = counts.tolist()
    assert c[1] == n and c[2] == lib.DRT_TRACE_OVERFLOW_PATHS
    assert set(k2.tolist()) <= set(k0[:n].tolist())
    # survivor-queue overflow: status bit 1
    lib.call("drt_trace_paths_compact_async", mesh.handle().h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx),
             rx.shape[0], C.byref(cands), 4, cap_p, ptr(k1), ptr(v1), ptr(o1), ptr(counts), ptr(ws), nb, stream())
    torch.cuda.synchronize()
    c = counts.tolist()
    assert c[0] > 4 and (c[2] & lib.DRT_TRACE_OVERFLOW_SURVIVORS)


def test_capture_and_replay_trace_forward_and_vjp():
    """The north-star mode under a HIP graph: async compact trace + VJP over the FIXED capacity are
    captured once and replayed; keys / vertices / objects are bit-identical to the synchronous entry
    point, the gradient equals the one of the synchronous pipeline."""
    from differt_amd._tensors import ptr, stream

    C, lib, mesh, tx, rx, params, cands, order = _trace_setup()
    L = lib.load()
    cap_s, cap_p = 1 << 16, 256
    nb = L.drt_trace_compact_workspace_size(cap_s, cap_p)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    keys = torch.empty(cap_p, dtype=torch.int64, device="cuda")
    verts = torch.empty((cap_p, order + 2, 3), dtype=torch.float32, device="cuda")
    objs = torch.empty((cap_p, order + 2), dtype=torch.int32, device="cuda")
    counts = torch.zeros(4, dtype=torch.int64, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    cot = torch.randn((cap_p, order + 2, 3), device="cuda", generator=gen)
    gtx, grx = torch.zeros_like(tx), torch.zeros_like(rx)
    gmv = torch.zeros_like(mesh.vertices)
    h = mesh.handle().h

    def launch():
        gtx.zero_(); grx.zero_(); gmv.zero_()
        lib.call("drt_trace_paths_compact_async", h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx), rx.shape[0],
                 C.byref(cands), cap_s, cap_p, ptr(keys), ptr(verts), ptr(objs), ptr(counts), ptr(ws), nb, stream())
        lib.call("drt_trace_paths_vjp", h, ptr(tx), tx.shape[0], ptr(rx), rx.shape[0], C.byref(cands), ptr(keys),
                 ptr(cot), cap_p, ptr(gtx), ptr(grx), ptr(gmv), stream())

    # reference: synchronous entry point + VJP over exactly the valid rows
    k0 = torch.empty(cap_p, dtype=torch.int64, device="cuda")
    v0 = torch.empty_like(verts)
    o0 = torch.empty_like(objs)
    nv = C.c_int64(0)
    lib.call("drt_trace_paths_compact", h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx), rx.shape[0],
             C.byref(cands), cap_s, cap_p, ptr(k0), ptr(v0), ptr(o0), C.byref(nv), ptr(ws), nb, stream())
    n = int(nv.value)
    assert 0 < n < cap_p
    rtx, rrx, rmv = torch.zeros_like(tx), torch.zeros_like(rx), torch.zeros_like(mesh.vertices)
    lib.call("drt_trace_paths_vjp", h, ptr(tx), tx.shape[0], ptr(rx), rx.shape[0], C.byref(cands), ptr(k0),
             ptr(cot), n, ptr(rtx), ptr(rrx), ptr(rmv), stream())
    torch.cuda.synchronize()

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch()  # warm-up outside the capture (module load, rocPRIM kernels)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for rep in range(3):
        for x in (keys, verts, objs, counts):
            x.fill_(13)
        g.replay()
        torch.cuda.synchronize()
        assert counts.tolist()[1] == n and counts.tolist()[2] == 0
        assert torch.equal(keys[:n], k0[:n]) and torch.equal(objs[:n], o0[:n])
        assert torch.equal(verts[:n].view(torch.int32), v0[:n].view(torch.int32))
        assert bool((keys[n:] == -1).all())
        for got, exp in ((gtx, rtx), (grx, rrx), (gmv, rmv)):
            scale = float(exp.abs().max()) + 1e-30
            assert float((got - exp).abs().max()) <= 1e-5 * scale  # float atomics: order may differ
    assert float(rtx.abs().max()) > 0


# ------------------------------------------------------------------ drt_trace_paths_beam_async ----
def _beam_city(assume_quads=False, boxes=30, ntx=3, nrx=40, seed=3):
    import numpy as np

    import differt_amd.geometry as G
    import synthetic_scenes as S

    V, Tr, c, h = S.manhattan(boxes, seed=seed)
    ext = float(np.abs(V[:, :2]).max()) + 10  # a ground quad: reflections off the street exist at every order
    gv = np.array([[-ext, -ext, 0], [ext, -ext, 0], [ext, ext, 0], [-ext, ext, 0]], np.float32)
    Tr = np.concatenate((Tr, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V)))
    V = np.concatenate((V, gv))
    _, rx = S.manhattan_tx_rx(c, h, 1, nrx, seed=seed + 1)
    _, tx = S.manhattan_tx_rx(c, h, 1, ntx, seed=seed + 2)  # transmitters above street crossings too (not inside a box)
    tx[:, 2] = np.linspace(3.0, 25.0, ntx, dtype=np.float32)
    mesh = G.Mesh(V, Tr, assume_quads=assume_quads)
    return G, mesh, torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda")


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["pairs", "triangles", "quads"])
def test_beam_async_equals_sync(order, kind):
    """One fixed-capacity pass without read-backs (drt_trace_paths_beam_async) == the slicing, synchronising entry
    point: same count, keys, objects, vertex bits; padding rows behind; status word 0."""
    G, mesh, tx, rx = _beam_city(assume_quads=(kind == "quads"), boxes=30 if order < 3 else 14)
    scene = G.Scene(tx, rx, mesh)
    tracer = G.ExhaustivePathTracer(accel="bvh")
    pairs = kind != "triangles"
    ref = tracer.trace_beam_pruned(scene, order, pairs=pairs)
    assert tracer.last_beam_stats["pair_mode"] == (kind == "pairs" and order > 0)
    n = ref.objects.shape[0]
    cap = 4096
    out = tracer.trace_beam_pruned_static(scene, order, max_paths=cap, pairs=pairs)
    torch.cuda.synchronize()
    c = out["counts"].tolist()
    assert c[1] == n and c[2] == 0, c
    assert torch.equal(out["keys"][:n], ref.keys) and torch.equal(out["objects"][:n], ref.objects)
    assert torch.equal(out["vertices"][:n].view(torch.int32), ref.vertices.view(torch.int32))
    assert bool((out["keys"][n:] == -1).all()) and bool((out["objects"][n:] == -1).all())
    assert bool((out["vertices"][n:] == 0).all())
    assert order == 0 or n > 0


def test_beam_async_capture_and_replay_with_moved_transmitters():
    """Captured once in a HIP graph, replayed after the transmitters moved IN PLACE (the buffers an XLA executable
    would re-use): every replay equals a fresh synchronous call on the new positions -- list sizes, the error unit and
    the receivers' box all live on the device."""
    G, mesh, tx, rx = _beam_city(boxes=40, ntx=4, nrx=64)
    tracer = G.ExhaustivePathTracer(accel="bvh")
    order, cap = 2, 8192
    scene = G.Scene(tx, rx, mesh)
    out = tracer.trace_beam_pruned_static(scene, order, max_paths=cap)  # builds clusters / LBVH, allocates buffers
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        tracer.trace_beam_pruned_static(scene, order, max_paths=cap, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tracer.trace_beam_pruned_static(scene, order, max_paths=cap, out=out)
    gen = torch.Generator(device="cuda").manual_seed(11)
    seen = set()
    for rep in range(4):
        if rep:
            tx.add_(torch.randn(tx.shape, device="cuda", generator=gen) * (3.0 if rep < 3 else 300.0))  # (rep 3: the magnitude, hence the unit, changes)
        for k in ("keys", "vertices", "objects", "counts"):
            out[k].fill_(13)
        g.replay()
        torch.cuda.synchronize()
        ref = tracer.trace_beam_pruned(G.Scene(tx.clone(), rx, mesh), order)
        n = ref.objects.shape[0]
        c = out["counts"].tolist()
        assert c[1] == n and c[2] == 0, (rep, c, n)
        assert torch.equal(out["keys"][:n], ref.keys) and torch.equal(out["objects"][:n], ref.objects)
        assert torch.equal(out["vertices"][:n].view(torch.int32), ref.vertices.view(torch.int32))
        assert bool((out["keys"][n:] == -1).all())
        seen.add(n)
    assert len(seen) > 1  # the replays really saw different problems


def test_beam_async_graph_replays_with_sorts_beyond_2e20_keys():
    """Regression (round 4): rocPRIM's radix sort takes its one-sweep algorithm above 2^20 items and resets its buffers
    with hipMemsetAsync -- memset NODES under capture, which replay once and then fill garbage on this ROCm: the second
    replay of a captured drt_trace_paths_beam_async on a 200 000-triangle mesh died with a memory aperture violation.
    Every sort of a capturable entry point now uses the merge-sort configuration (csrc/sort_safe.hpp).  Here: a row
    capacity of 2^23 (2^21 pair rows > 2^20 keys in the captured sort), five replays, each equal to the synchronous call."""
    G, mesh, tx, rx = _beam_city(boxes=600, ntx=2, nrx=48, seed=5)
    tracer = G.ExhaustivePathTracer(accel="bvh")
    order, cap = 2, 4096
    scene = G.Scene(tx, rx, mesh)
    caps = {"max_records": 1 << 22, "max_rows": 1 << 23, "max_survivors": 1 << 21}
    out = tracer.trace_beam_pruned_static(scene, order, max_paths=cap, **caps)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        tracer.trace_beam_pruned_static(scene, order, max_paths=cap, out=out, **caps)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tracer.trace_beam_pruned_static(scene, order, max_paths=cap, out=out, **caps)
    ref = tracer.trace_beam_pruned(scene, order)
    n = ref.objects.shape[0]
    assert n > 0
    for rep in range(5):
        for k in ("keys", "vertices", "objects", "counts"):
            out[k].fill_(13)
        g.replay()
        torch.cuda.synchronize()
        c = out["counts"].tolist()
        assert c[1] == n and c[2] == 0, (rep, c, n)
        assert torch.equal(out["keys"][:n], ref.keys) and torch.equal(out["objects"][:n], ref.objects)
        assert torch.equal(out["vertices"][:n].view(torch.int32), ref.vertices.view(torch.int32))


def test_beam_async_reports_overflow_instead_of_slicing():
    from differt_amd import _lib

    G, mesh, tx, rx = _beam_city(boxes=40, ntx=4, nrx=64)
    scene = G.Scene(tx, rx, mesh)
    tracer = G.ExhaustivePathTracer()
    ref = tracer.trace_beam_pruned(scene, 2)
    have = set(ref.keys.tolist())
    for kw, bit in (({"max_records": 1024}, _lib.DRT_BEAM_OVERFLOW_RECORDS), ({"max_rows": 2048}, _lib.DRT_BEAM_OVERFLOW_ROWS),
                    ({"max_survivors": 8}, _lib.DRT_TRACE_OVERFLOW_SURVIVORS)):
        out = tracer.trace_beam_pruned_static(scene, 2, max_paths=4096, **kw)
        torch.cuda.synchronize()
        c = out["counts"].tolist()
        assert c[2] & bit, (kw, c)
        got = [k for k in out["keys"].tolist() if k >= 0]
        assert set(got) <= have  # what is written is valid, only incomplete
    out = tracer.trace_beam_pruned_static(scene, 2, max_paths=4)
    torch.cuda.synchronize()
    assert out["counts"].tolist()[2] & _lib.DRT_TRACE_OVERFLOW_PATHS and out["counts"].tolist()[1] == len(have)
