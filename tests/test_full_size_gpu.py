"""Parity at BASELINE.json's FULL sizes (configs[1..4]) through size-independent properties, plus
oracle spot checks on samples the oracle finishes in seconds.

* sampled rows / paths are re-derived by the CPU oracle bit-exactly;
* consistency between independent kernels (dense MT vs any-hit vs first-hit);
* partition properties (a batch split in shards gives the concatenation; rank windows tile the
  candidate space);
* every valid path the GPU reports is re-validated by the oracle, and random candidates it
  rejected are rejected by the oracle.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc
import synthetic_scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _cfg2(R, T, seed=1234):
    rng = np.random.default_rng(seed)
    o = (rng.uniform(-1, 1, (R, 3)) * 50).astype(np.float32)
    d = (rng.uniform(-1, 1, (R, 3)) * 50).astype(np.float32) - o
    c = (rng.uniform(-1, 1, (T, 1, 3)) * 50).astype(np.float32)
    e = (rng.normal(size=(T, 2, 3)) * 2).astype(np.float32)
    return o, d, np.concatenate([c, c + e[:, :1], c + e[:, 1:]], axis=1).astype(np.float32)


def test_cfg2_full_size(G):
    """configs[1] at bench size: 65 536 rays x 10 000 triangles (6.5e8 tests)."""
    R, T = 65536, 10000
    o, d, tv = _cfg2(R, T)
    to, td, ttv = (torch.as_tensor(x, device="cuda") for x in (o, d, tv))
    t, hit = G.ray_intersect_triangle(to[:, None, :], td[:, None, :], ttv)
    assert tuple(t.shape) == (R, T)
    # (a) sampled rows == oracle, bit for bit
    rows = np.random.default_rng(0).choice(R, 192, replace=False)
    et, eh = orc.ray_intersect_triangle_dense(o[rows], d[rows], tv)
    np.testing.assert_array_equal(_np(hit[rows]), eh)
    np.testing.assert_array_equal(_bits(_np(t[rows])), _bits(et))
    assert eh.sum() > 100
    # (b) any-hit kernel == OR over the dense row
    thr = np.float32(1.0) - np.float32(100.0 * np.finfo(np.float32).eps)
    blocked = G.ray_intersect_any_triangle(to, td, ttv)
    torch.testing.assert_close(blocked, (hit & (t < float(thr))).any(dim=1))
    # (c) first-hit kernel == masked argmin of the dense row (one 512-tile semantics checked by the
    #     unit tests; here: same t, and the index is a minimiser)
    idx, tmin = G.first_triangle_hit_by_ray(to, td, ttv)
    tm = torch.where(hit, t, torch.full_like(t, float("inf")))
    torch.testing.assert_close(tmin, tm.min(dim=1).values, rtol=0, atol=0)
    has = idx >= 0
    assert bool((has == torch.isfinite(tmin)).all())
    picked = tm[torch.nonzero(has).squeeze(1), idx[has].long()]
    torch.testing.assert_close(picked, tmin[has], rtol=0, atol=0)
    # (d) sharding the ray axis = concatenation (the multi-GPU decomposition of bench.py)
    t2, h2 = G.ray_intersect_triangle(to[R // 2:, None, :], td[R // 2:, None, :], ttv)
    assert torch.equal(t2, t[R // 2:]) and torch.equal(h2, hit[R // 2:])
    # (e) idempotence: a second launch is bit-identical (no data race in the store pattern)
    t3, h3 = G.ray_intersect_triangle(to[:, None, :], td[:, None, :], ttv)
    assert torch.equal(t3.view(torch.int32), t.view(torch.int32)) and torch.equal(h3, hit)


@pytest.fixture(scope="module")
def manhattan():
    V, Tr, centres, heights = S.manhattan(1000)
    tx, rx = S.manhattan_tx_rx(centres, heights, 16, 64)
    return V, Tr, tx, rx


def _oracle_revalidate(V, Tr, tx, rx, paths, count, order, n_nodes, rank_lo=0):
    """Each reported path: the oracle, given ONLY that (tx, rx, candidate), says valid and returns the
    same vertices."""
    keys = _np(paths.keys)
    objs = _np(paths.objects)
    verts = _np(paths.vertices)
    nrx = rx.shape[0]
    for k, ob, vv in zip(keys, objs, verts):
        pair, row = divmod(int(k), count)
        it, ir = divmod(pair, nrx)
        assert ob[0] == it and ob[-1] == ir
        cand = np.asarray([ob[1:-1]], np.int32)
        # the candidate is the one the lexicographic rank denotes (GPU unranking == oracle odometer)
        o = orc.trace_path_candidates(V, Tr, tx[it: it + 1], rx[ir: ir + 1], cand)
        assert o["mask"].reshape(-1)[0], (k, ob)
        np.testing.assert_array_equal(_bits(o["vertices"].reshape(order + 2, 3)), _bits(vv))
    return keys


def test_cfg3_full_size(G, manhattan):
    """configs[2]: 16 TX x 64 RX, 10k triangles, order 2, ALL 99 990 000 candidates per pair
    (1.02e11 path candidates), fwd + grad(TX)."""
    V, Tr, tx, rx = manhattan
    n, order = Tr.shape[0], 2
    total = n * (n - 1)
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    scene = G.Scene(txg, rx, G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer()
    paths = tracer.trace_rank_range_literal(scene, order, max_survivors=1 << 24)
    nv = paths.objects.shape[0]
    assert nv >= 10
    keys = _oracle_revalidate(V, Tr, tx, rx, paths, total, order, n)
    assert (np.diff(keys) > 0).all()  # sorted, unique: the masked_vertices order
    # rank r of pair p <-> candidate (i, j): closed form == objects
    objs = _np(paths.objects)
    row = keys % total
    i = row // (n - 1)
    dd = row % (n - 1)
    np.testing.assert_array_equal(objs[:, 1], i)
    np.testing.assert_array_equal(objs[:, 2], dd + (dd >= i))
    # partition: 7 uneven rank windows tile the candidate space
    from differt_amd.distributed import globalize_keys, shard_interval

    got = []
    for r in range(7):
        lo, hi = shard_interval(total, 7, r)
        w = tracer.trace_rank_range_literal(scene, order, lo, hi, max_survivors=1 << 22)
        got.append(_np(globalize_keys(w.keys, hi - lo, lo, total)))
    np.testing.assert_array_equal(np.sort(np.concatenate(got)), keys)
    # rejected candidates: random ones and near-misses (one mirror swapped) are rejected by the oracle too
    rng = np.random.default_rng(1)
    valid = set(keys.tolist())
    probes = []
    for k in keys[:8]:
        pair, r = divmod(int(k), total)
        for delta in (-2, -1, 1, 2, n - 1, -(n - 1)):
            if 0 <= r + delta < total:
                probes.append(pair * total + r + delta)
    probes += rng.integers(0, 1024 * total, 64).tolist()
    for k in probes:
        pair, r = divmod(int(k), total)
        it, ir = divmod(pair, rx.shape[0])
        ci, dj = divmod(r, n - 1)
        cand = np.asarray([[ci, dj + (dj >= ci)]], np.int32)
        o = orc.trace_path_candidates(V, Tr, tx[it: it + 1], rx[ir: ir + 1], cand)
        assert bool(o["mask"].reshape(-1)[0]) == (k in valid), (k, cand)
    # completeness at FULL size, by an independent route: the geometric (beam) pruning reaches the valid paths
    # through necessary conditions on pyramids / mirror sides instead of evaluating the 1.02e11 candidates; a
    # path that the filter kernel wrongly dropped at full scale would show up here (and vice versa)
    bp = tracer.trace_beam_pruned(scene, order)
    assert torch.equal(bp.objects, paths.objects)
    assert torch.equal(bp.vertices.detach().view(torch.int32), paths.vertices.detach().view(torch.int32))
    assert tracer.last_beam_stats["rows"] < 1e-4 * 1024 * total
    # triangle-block sharding of the occlusion stage, 8 blocks emulated on one GPU: a path is valid iff no block
    # occludes it -> the intersection of the per-block results is the unsharded result
    from differt_amd.distributed import trace_rank_range_triangle_sharded

    lo, hi = shard_interval(total, 7, 3)  # one rank window is enough (the geometric stage is the same code)
    ref_keys = set(_np(tracer.trace_rank_range_literal(scene, order, lo, hi, max_survivors=1 << 22).keys).tolist())
    common = None
    for b in range(8):
        t0, t1 = shard_interval(n, 8, b)
        part = trace_rank_range_triangle_sharded(tracer, scene, order, lo, hi, tri_lo=t0, tri_hi=t1,
                                                 max_survivors=1 << 22)
        ks = set(_np(part.keys).tolist())
        common = ks if common is None else (common & ks)
        assert ref_keys <= ks  # a block can only occlude less than the whole mesh
    assert common == ref_keys and len(ref_keys) > 0
    # gradient of the total valid path length w.r.t. TX: finite, and equal to float64 autograd
    from oracle import torch_ref

    torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
    g = _np(txg.grad)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    vt = torch.tensor(V, dtype=torch.float64)
    txt = torch.tensor(tx, dtype=torch.float64, requires_grad=True)
    rxt = torch.tensor(rx, dtype=torch.float64)
    loss = 0.0
    for ob in objs:
        full = torch_ref.trace_vertices(vt, torch.tensor(Tr, dtype=torch.long), txt[ob[0]: ob[0] + 1],
                                        rxt[ob[-1]: ob[-1] + 1], torch.tensor([ob[1:-1]], dtype=torch.long))
        loss = loss + torch.sqrt((torch.diff(full.reshape(order + 2, 3), dim=0) ** 2).sum(-1)).sum()
    loss.backward()
    e = txt.grad.numpy()
    np.testing.assert_allclose(g, e, rtol=1e-5, atol=1e-5 * np.abs(e).max())


def test_cfg4_order3_rank_window(G, manhattan):
    """configs[3]: same scene, order 3 (9.998e11 candidates per pair): rank windows of 2^22 ranks per
    pair, GPU-unranked; window tiling + oracle re-validation."""
    V, Tr, tx, rx = manhattan
    n, order = Tr.shape[0], 3
    scene = G.Scene(tx[:4], rx[:8], G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer()
    # start inside the space, across a first-mirror boundary: c0 = 4711 starts at 4711 * (n-1)^2
    lo = 4711 * (n - 1) ** 2 - (1 << 21)
    cnt = 1 << 22
    whole = tracer.trace_rank_range_literal(scene, order, lo, lo + cnt, max_survivors=1 << 22)
    a = tracer.trace_rank_range_literal(scene, order, lo, lo + cnt // 2, max_survivors=1 << 22)
    b = tracer.trace_rank_range_literal(scene, order, lo + cnt // 2, lo + cnt, max_survivors=1 << 22)
    from differt_amd.distributed import globalize_keys

    tot = n * (n - 1) ** 2
    gk = np.sort(np.concatenate([_np(globalize_keys(a.keys, cnt // 2, lo, tot)),
                                 _np(globalize_keys(b.keys, cnt // 2, lo + cnt // 2, tot))]))
    np.testing.assert_array_equal(gk, _np(globalize_keys(whole.keys, cnt, lo, tot)))
    # unranked candidates == host unranking == oracle's closed form, on sampled rows
    rows = np.random.default_rng(2).integers(0, cnt, 2000)
    table = torch.empty((cnt, order), dtype=torch.int32, device="cuda")
    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream

    _lib.call("drt_candidates_fill", n, order, lo, lo + cnt, None, 1, ptr(table), stream())
    host = G.CompleteGraph(n).all_paths_array(n, n + 1, order + 2, include_from_and_to=False,
                                              rank_lo=lo, rank_hi=lo + cnt)
    np.testing.assert_array_equal(_np(table)[rows], host[rows].astype(np.int32))
    assert (_np(table)[:, 1:] != _np(table)[:, :-1]).all()
    # oracle spot check of the mask on a sub-window (dense oracle: 4 x 8 pairs x 3000 candidates)
    sub = 3000
    cand = host[:sub].astype(np.int32)
    o = orc.trace_path_candidates(V, Tr, tx[:4], rx[:8], cand)
    w = tracer.trace_rank_range_literal(scene, order, lo, lo + sub)
    np.testing.assert_array_equal(_np(w.keys), np.flatnonzero(o["mask"].reshape(-1)))


def test_cfg4_full_coverage_beam_pruned(G, manhattan):
    """configs[3] at FULL size: order 3 over all 1.02e15 candidates through the conservatively pruned search
    (DESIGN.md section 9).  No exhaustive run can cross-check it; what can be checked here: every returned path
    is valid and bit-identical for the oracle given only its candidate (soundness), the keys are the closed form
    of the objects in strictly increasing order, the independent sampled per-pair search finds no path the
    pruned search lacks (it may find fewer), and two prefix shards partition the result."""
    V, Tr, tx, rx = manhattan
    n, order = Tr.shape[0], 3
    scene = G.Scene(tx, rx, G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer(accel="bvh")
    bp = tracer.trace_beam_pruned(scene, order)
    st = dict(tracer.last_beam_stats)
    assert bp.objects.shape[0] >= 50 and st["rows"] < 1e-6 * 1024 * n * (n - 1) ** 2
    _oracle_revalidate(V, Tr, tx, rx, bp, n ** order, order, n)
    o = bp.objects.long()
    keys = (o[:, 0] * rx.shape[0] + o[:, 4]) * n ** 3 + o[:, 1] * n * n + o[:, 2] * n + o[:, 3]
    assert torch.equal(bp.keys, keys) and bool((keys[1:] > keys[:-1]).all())
    sampled = G.HybridPathTracer(num_rays=100_000, accel="bvh", sample_triangles=True).trace_pairs(scene, order)
    have = set(map(tuple, _np(bp.objects).tolist()))
    assert set(map(tuple, _np(sampled.objects).tolist())) <= have
    parts = [tracer.trace_beam_pruned(scene, order, prefix_shard=(r, 2)) for r in range(2)]
    merged = torch.sort(torch.cat([p.keys for p in parts])).values
    assert torch.equal(merged, bp.keys)


def _exhaustive_pairs_module():
    import importlib.util
    from pathlib import Path

    # tests/exhaustive_pairs_check.py (scratch/exhaustive_pairs.py is its command line)
    spec = importlib.util.spec_from_file_location("exhaustive_pairs_check", Path(__file__).resolve().parent / "exhaustive_pairs_check.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cfg4_beam_equals_exhaustive_on_a_whole_pair_space(G, manhattan):
    """configs[3] completeness AT ITS OWN SIZE: every one of the 9.998e11 order-3 candidates of
    one (tx, rx) pair -- the pair with the most paths -- through the exhaustive tracer (reference enumeration
    _solvers.py:803-848), against the rows of the full 16 x 64 pruned search that belong to the pair, at kappa = 64
    and kappa = 1: same objects, same vertex bits.  ~12 s; scratch/exhaustive_pairs.py runs more pairs (record:
    profiles/r04/stress/exhaustive_pairs.json)."""
    V, Tr, tx, rx = manhattan
    rec = _exhaustive_pairs_module().check_config("configs[3]", V, Tr, tx, rx, 3, 1, (64.0, 1.0), 64, 1 << 24)
    assert rec["all_equal"] and rec["checked_pairs"] == 1
    assert rec["pairs"][0]["candidates"] == 10000 * 9999 ** 2 and rec["pairs"][0]["exhaustive_valid_paths"] >= 1


def test_cfg5_beam_equals_exhaustive_on_whole_pair_spaces(G):
    """configs[4] (200 000 triangles, order 2): all 4.0e10 candidates of 3 pairs (most paths / grazing-heavy / no
    path) against the pruned search, kappa = 64 and 1."""
    V, Tr, tx, rx = S.cfg5_scene()
    rec = _exhaustive_pairs_module().check_config("configs[4]", V, Tr, tx, rx, 2, 3, (64.0, 1.0), 4, 1 << 24)
    assert rec["all_equal"] and rec["checked_pairs"] == 3
    assert sum(p["exhaustive_valid_paths"] for p in rec["pairs"]) >= 1


def test_cfg5_full_coverage_beam_pruned(G):
    """configs[4] at FULL size (1 TX x 1024 RX, 200 000 triangles, order 2, 4.1e13 candidates) through the pruned
    search with the clustered expansion / receiver stage: every path re-validated by the oracle against the whole
    mesh, the plain receiver stage gives the same rows, fwd + grad finite."""
    V, Tr, tx, rx = S.cfg5_scene()
    n, order = Tr.shape[0], 2
    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    scene = G.Scene(txg, torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer(accel="bvh")
    bp = tracer.trace_beam_pruned(scene, order)
    rows = tracer.last_beam_stats["rows"]
    assert bp.objects.shape[0] >= 50 and rows < 1e-5 * 1024 * n * (n - 1)
    _oracle_revalidate(V, Tr, tx, rx, bp, n ** order, order, n)
    plain = tracer.trace_beam_pruned(scene, order, emit="plain")
    assert tracer.last_beam_stats["rows"] == rows and torch.equal(plain.keys, bp.keys)
    torch.sqrt((torch.diff(bp.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
    assert bool(torch.isfinite(txg.grad).all()) and float(txg.grad.abs().max()) > 0


def test_cfg5_200k_triangles(G):
    """configs[4] scene: 20 000 boxes = 200 000 triangles, 1 TX x 1024 RX grid, order 2: first-hit /
    any-hit over the full mesh vs the oracle on sampled rays, and a rank window of the tracer."""
    V, Tr, centres, heights = S.manhattan(20000)
    assert Tr.shape[0] == 200000
    tx, _ = S.manhattan_tx_rx(centres, heights, 1, 1)
    g = np.linspace(-400, 400, 32, dtype=np.float32)
    rx = np.stack(np.meshgrid(g + 20, g + 20, indexing="xy"), -1).reshape(-1, 2)
    rx = np.column_stack((rx, np.full(len(rx), 1.5, np.float32))).astype(np.float32)
    mesh = G.Mesh(V, Tr)
    tv = orc.triangle_vertices(V, Tr)
    # rays TX -> every RX: first hit and occlusion over 200k triangles
    o = np.broadcast_to(tx, rx.shape).copy()
    d = rx - o
    idx, t = mesh.first_triangle_hit_by_ray(o, d)
    blocked = mesh.ray_intersect_any_triangle(o, d)
    sel = np.random.default_rng(3).choice(len(rx), 48, replace=False)
    ei, et = orc.first_triangle_hit_by_ray(o[sel], d[sel], tv)
    np.testing.assert_array_equal(_np(idx)[sel], ei)
    np.testing.assert_array_equal(_np(t)[sel], et)
    np.testing.assert_array_equal(_np(blocked)[sel], orc.ray_intersect_any_triangle(o[sel], d[sel], tv))
    # triangle-block sharding (configs[4]: 8 blocks of 25 000 triangles): per-block packed keys with
    # GLOBAL tile ids, MIN-combined (what the RCCL all-reduce does), decoded == the unsharded operator,
    # indices included, for tiles that straddle block boundaries (25 000 % 512 != 0) and with ties
    from differt_amd.distributed import first_triangle_hit_by_ray_sharded

    ttv = torch.as_tensor(tv, device="cuda")
    ttv[[24999, 25000, 150000]] = ttv[12345].clone()  # exact duplicates in different blocks
    ref_i, ref_t = G.first_triangle_hit_by_ray(o, d, ttv)
    flip = torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda")
    combined = None
    for b in range(8):
        _, _, keys = first_triangle_hit_by_ray_sharded(o, d, ttv[b * 25000:(b + 1) * 25000], b * 25000, 200000)
        signed = torch.bitwise_xor(keys, flip)
        combined = signed if combined is None else torch.minimum(combined, signed)
    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream

    keys = torch.bitwise_xor(combined, flip).contiguous()
    si = torch.empty(len(rx), dtype=torch.int32, device="cuda")
    st = torch.empty(len(rx), dtype=torch.float32, device="cuda")
    _lib.call("drt_first_hit_finalize", ptr(keys), len(rx), 200000, 512, ptr(si), ptr(st), stream())
    assert torch.equal(si, ref_i) and torch.equal(st, ref_t)
    # a ray aimed at the duplicated triangle: the later 512-tile wins (index 150000)
    cen = ttv[12345].mean(dim=0)
    oo = (cen + torch.tensor([0.0, 0.0, 500.0], device="cuda"))[None]
    dd = (cen - oo[0])[None] * 2
    ti, _ = G.first_triangle_hit_by_ray(oo, dd, ttv)
    combined = None
    for b in range(8):
        _, _, k = first_triangle_hit_by_ray_sharded(oo, dd, ttv[b * 25000:(b + 1) * 25000], b * 25000, 200000)
        sk = torch.bitwise_xor(k, flip)
        combined = sk if combined is None else torch.minimum(combined, sk)
    k = torch.bitwise_xor(combined, flip).contiguous()
    i1 = torch.empty(1, dtype=torch.int32, device="cuda")
    t1 = torch.empty(1, dtype=torch.float32, device="cuda")
    _lib.call("drt_first_hit_finalize", ptr(k), 1, 200000, 512, ptr(i1), ptr(t1), stream())
    assert int(i1) == int(ti)
    # a rank window of the order-2 tracer on the big mesh, re-validated by the oracle
    scene = G.Scene(tx, rx[:64], mesh)
    n = 200000
    lo = 123456 * (n - 1)
    w = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, lo, lo + 3 * (n - 1), max_survivors=1 << 22)
    _oracle_revalidate(V, Tr, tx, rx[:64], w, 3 * (n - 1), 2, n)


def test_sharded_driver_single_process(G, manhattan):
    """differt_amd.distributed.trace_rank_range_sharded with world_size 1 == trace_rank_range, and a
    manual 3-way emulation of the sharding (what N ranks would each compute) == the whole."""
    from differt_amd.distributed import gather_paths, globalize_keys, shard_interval, trace_rank_range_sharded

    V, Tr, tx, rx = manhattan
    scene = G.Scene(tx[:2], rx[:4], G.Mesh(V, Tr))
    tracer = G.ExhaustivePathTracer()
    lo, hi = 30_000_000, 60_000_000
    keys, verts, objs = trace_rank_range_sharded(tracer, scene, 2, lo, hi, max_survivors=1 << 22)
    whole = tracer.trace_rank_range_literal(scene, 2, lo, hi, max_survivors=1 << 22)
    assert torch.equal(keys, whole.keys) and torch.equal(verts, whole.vertices) and torch.equal(objs, whole.objects)
    parts = []
    for r in range(3):
        a, b = shard_interval(hi - lo, 3, r)
        p = tracer.trace_rank_range_literal(scene, 2, lo + a, lo + b, max_survivors=1 << 22)
        parts.append((globalize_keys(p.keys, b - a, a, hi - lo), p.vertices, p.objects))
    k = torch.cat([p[0] for p in parts])
    perm = torch.argsort(k, stable=True)
    assert torch.equal(k[perm], whole.keys)
    assert torch.equal(torch.cat([p[1] for p in parts])[perm], whole.vertices)
