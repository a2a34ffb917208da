"""The dense-layout tracer -- the operator under the reference's unchanged ``Scene.trace_paths(order, chunk_size=...)``
/ ``solver.trace_path_candidates`` signatures (reference _scene.py:735-764, _solvers.py:499-770, 850-957) -- against the
oracle, bit for bit, on every store path of ``trace_dense_kernel`` (csrc/trace_dense.hip):

* 16-byte line-aligned stores (rows per chunk a multiple of 16), 16-byte stores at a line offset (multiple of 4, 8),
  4-byte coalesced stores (odd row counts), partial last waves / blocks, several (tx, rx);
* interaction types broadcast over (tx, rx) (_solvers.py:751-762), ``-1`` padded chunks (:912-918);
* the GPU-filled chunk iterator == the host enumeration (graph.rs order), chunk by chunk.
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


def _city(G, boxes=12, ntx=2, nrx=3, assume_quads=False, mask=None):
    V, Tr, centres, heights = S.manhattan(boxes)
    tx, rx = S.manhattan_tx_rx(centres, heights, ntx, nrx)
    # a low transmitter too: paths with reflections off the walls exist at street level
    tx[0] = (rx[0] + np.array([3.0, 2.0, 4.0], np.float32))
    return V, Tr, tx, rx, G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask, assume_quads=assume_quads))


def _check(got, o, types=None):
    np.testing.assert_array_equal(_np(got.mask).astype(np.uint8), o["mask"].astype(np.uint8))
    np.testing.assert_array_equal(_np(got.objects), o["objects"])
    np.testing.assert_array_equal(_bits(_np(got.vertices)), _bits(o["vertices"]))
    it = _np(got.interaction_types)
    assert it.shape == o["objects"].shape[:-1] + (o["objects"].shape[-1] - 2,)
    if types is None:
        assert (it == 0).all()
    else:
        np.testing.assert_array_equal(it, np.broadcast_to(types, it.shape))


@pytest.mark.parametrize("order,count", [
    (0, 1), (1, 120), (1, 117),
    (2, 4096), (2, 1000), (2, 1001), (2, 1003), (2, 1028), (2, 63), (2, 321),
    (3, 2048), (3, 1002), (3, 1004), (3, 999), (4, 512), (4, 257),
])
@pytest.mark.parametrize("assume_quads", [False, True])
def test_dense_every_store_path_vs_oracle(G, rng, order, count, assume_quads):
    V, Tr, tx, rx, scene = _city(G, assume_quads=assume_quads)
    n = scene.mesh.num_primitives
    total = 1 if order == 0 else n * (n - 1) ** (order - 1)
    count = min(count, total)
    if order == 0:
        cand = np.zeros((1, 0), np.int32)
    else:
        # a window of the lexicographic space + rows around the transmitter's own building (valid paths live there)
        lo = int(rng.integers(0, min(total - count, 200_000) + 1))  # bounded: the oracle enumerates from rank 0
        it = orc.CompleteGraphIter(n, n, n + 1, order + 2, False)
        allc = it.collect_array(min(total, lo + count)).astype(np.int32)[lo:lo + count]
        cand = allc * (2 if assume_quads else 1)
    types = rng.integers(0, 3, size=cand.shape).astype(np.int32)
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand, assume_quads=assume_quads)
    got = scene.trace_paths(path_candidates=cand)
    _check(got, o)
    tr = G.ExhaustivePathTracer()
    got2 = tr.trace_path_candidates(scene, cand, types)
    _check(got2, o, types)


def test_dense_finds_paths_and_masks_blocked_ones(G):
    """The parity cases above must not be vacuous: a full order-2 space with valid AND blocked survivors."""
    V, Tr, tx, rx, scene = _city(G, boxes=12, ntx=3, nrx=4)
    n = scene.mesh.num_primitives
    cand = orc.generate_all_path_candidates(n, 2).astype(np.int32)
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand, return_diag=True)
    got = scene.trace_paths(2)
    _check(got, o)
    assert o["mask"].sum() >= 3
    # chunked through the reference call: same masks, every chunk a GPU-filled table
    for cs in (4096, 1000, 777):
        chunks = list(scene.trace_paths(2, chunk_size=cs))
        assert len(chunks) == -(-cand.shape[0] // cs)
        np.testing.assert_array_equal(np.concatenate([_np(c.mask) for c in chunks], axis=-1), o["mask"].astype(bool))
        np.testing.assert_array_equal(np.concatenate([_np(c.objects) for c in chunks], axis=-2), o["objects"])
        np.testing.assert_array_equal(_bits(np.concatenate([_np(c.vertices) for c in chunks], axis=-3)),
                                      _bits(o["vertices"]))


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("disconnect", [False, True])
def test_chunk_iterator_is_gpu_filled_and_equals_host_enumeration(G, rng, order, assume_quads, disconnect):
    """generate_path_candidates_chunks_iter (_solvers.py:850-934): lexicographic rows (graph.rs:400-470), x2 for quads
    (:842-843), inactive primitives skipped with disconnect_inactive_triangles (:820-827), -1 padding (:912-918)."""
    V, Tr, *_ = _city(G, boxes=3)
    mask = rng.random(Tr.shape[0]) > 0.3
    if assume_quads:
        mask[1::2] = mask[0::2]
    *_, scene = _city(G, boxes=3, assume_quads=assume_quads, mask=mask)
    tracer = G.ExhaustivePathTracer(disconnect_inactive_triangles=disconnect)
    n = scene.mesh.num_primitives
    if disconnect:
        pm = mask[0::2] & mask[1::2] if assume_quads else mask
        act = np.flatnonzero(pm)
        exp = act[orc.generate_all_path_candidates(len(act), order)]
    else:
        exp = orc.generate_all_path_candidates(n, order)
    exp = exp.astype(np.int32) * (2 if assume_quads else 1)
    for cs in (64, 1000, 37):
        it = tracer.generate_path_candidates_chunks_iter(scene, order, chunk_size=cs)
        assert len(it) == -(-exp.shape[0] // cs)
        chunks = list(it)
        assert all(c.is_cuda and c.dtype == torch.int32 for c, _ in chunks)
        np.testing.assert_array_equal(np.concatenate([_np(c) for c, _ in chunks]), exp)
        assert all((_np(t) == 0).all() and t.shape == c.shape for c, t in chunks)
        padded = list(tracer.generate_path_candidates_chunks_iter(scene, order, chunk_size=cs, pad_chunks=True))
        assert all(tuple(c.shape) == (cs, order) for c, _ in padded)
        flat = np.concatenate([_np(c) for c, _ in padded])
        np.testing.assert_array_equal(flat[:exp.shape[0]], exp)
        # the reference pads with -1 and THEN doubles quad ids (_solvers.py:912-925): -2 on a quad mesh, host path alike
        assert (flat[exp.shape[0]:] == (-2 if assume_quads else -1)).all()


def test_padded_chunk_rows_are_invalid_and_zero(G):
    V, Tr, tx, rx, scene = _city(G, boxes=3)
    tracer = G.ExhaustivePathTracer()
    n = scene.mesh.num_primitives
    total = n * (n - 1)
    cs = 256
    padded = list(tracer.generate_path_candidates_chunks_iter(scene, 2, chunk_size=cs, pad_chunks=True))
    last = tracer.trace_path_candidates(scene, *padded[-1])
    real = total - (len(padded) - 1) * cs
    assert 0 < real < cs
    assert not _np(last.mask)[..., real:].any()
    assert (_np(last.vertices)[..., real:, :, :] == 0).all()
    assert (_np(last.objects)[..., real:, 1:-1] == -1).all()
    o = orc.trace_path_candidates(V, Tr, tx, rx, _np(padded[-1][0])[:real])
    np.testing.assert_array_equal(_bits(_np(last.vertices)[..., :real, :, :]), _bits(o["vertices"]))
    np.testing.assert_array_equal(_np(last.mask)[..., :real], o["mask"].astype(bool))


def test_dense_raw_call_unaligned_output_pointers(G):
    """drt_trace_paths_dense_ex with output pointers that are only 4-byte aligned: the 4-byte store path, same bits."""
    import ctypes as C

    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream
    from differt_amd.geometry._solvers import _params, _table_candidates

    V, Tr, tx, rx, scene = _city(G, boxes=6)
    n = scene.mesh.num_primitives
    cand = orc.generate_all_path_candidates(n, 2).astype(np.int32)[:2048]
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand)
    dev = "cuda"
    table = torch.as_tensor(cand, device=dev)
    ntx, nrx, Cn, k = tx.shape[0], rx.shape[0], cand.shape[0], 2
    nv, no, nt, nm = ntx * nrx * Cn * (k + 2) * 3, ntx * nrx * Cn * (k + 2), ntx * nrx * Cn * k, ntx * nrx * Cn
    bv = torch.full((nv + 1,), 7.0, dtype=torch.float32, device=dev)
    bo = torch.full((no + 1,), 7, dtype=torch.int32, device=dev)
    bt = torch.full((nt + 1,), 7, dtype=torch.int32, device=dev)
    bm = torch.full((nm + 3,), 7, dtype=torch.uint8, device=dev)
    txd, rxd = torch.as_tensor(tx, device=dev), torch.as_tensor(rx, device=dev)
    lib = _lib.load()
    nbytes = lib.drt_trace_dense_workspace_size(ntx, nrx, Cn)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    params = _params(None, None, None)
    cands = _table_candidates(table)
    _lib.call("drt_trace_paths_dense_ex", scene.mesh.handle().h, C.byref(params), ptr(txd), ntx, ptr(rxd), nrx,
              C.byref(cands), None, C.c_void_p(bv.data_ptr() + 4), C.c_void_p(bo.data_ptr() + 4),
              C.c_void_p(bm.data_ptr() + 3), C.c_void_p(bt.data_ptr() + 4), ptr(ws), nbytes, stream())
    torch.cuda.synchronize()
    assert float(bv[0]) == 7.0 and int(bo[0]) == 7 and int(bt[0]) == 7 and (_np(bm[:3]) == 7).all()
    np.testing.assert_array_equal(_bits(_np(bv[1:])), _bits(o["vertices"]).reshape(-1))
    np.testing.assert_array_equal(_np(bo[1:]), o["objects"].reshape(-1))
    np.testing.assert_array_equal(_np(bm[3:]), o["mask"].reshape(-1).astype(np.uint8))
    assert (_np(bt[1:]) == 0).all()


# ------------------------------------------------------------------ stand-alone image method (staged rows) ----
@pytest.mark.parametrize("k", [1, 2, 3, 8])
@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000, 4099])
@pytest.mark.parametrize("shared", ["none", "from_to", "mirrors"])
def test_image_method_staged_rows_bit_exact(G, rng, k, B, shared):
    """Every wave shape of image_method_kernel (full / partial waves, 16-byte and 4-byte row paths, rows shared by the
    batch read in place: the reference harness passes from / to of shape [3], tests/benchmarks/fixtures.py:19-40)."""
    a = rng.normal(size=(3,) if shared == "from_to" else (B, 3)).astype(np.float32)
    b = rng.normal(size=(3,) if shared == "from_to" else (B, 3)).astype(np.float32)
    ms = (k, 3) if shared == "mirrors" else (B, k, 3)
    mv = rng.normal(size=ms).astype(np.float32)
    mn, _ = orc.normalize(rng.normal(size=ms).astype(np.float32))
    if shared == "mirrors":
        a, b = rng.normal(size=(B, 3)).astype(np.float32), rng.normal(size=(B, 3)).astype(np.float32)
    exp = orc.image_method(a, b, mv, mn)
    got = G.image_method(a, b, mv, mn)
    assert tuple(got.shape) == (B, k, 3)
    np.testing.assert_array_equal(_bits(_np(got)), _bits(np.broadcast_to(exp, (B, k, 3))))


@pytest.mark.parametrize("k", [1, 2, 5])
def test_image_method_vjp_with_shared_rows(G, rng, k):
    """Gradients w.r.t. inputs that are broadcast over the batch (read in place, stride 0) = the sum over the batch of
    the per-element gradients the dense call returns (parity of those with float64 autograd:
    tests/test_trace_gpu.py::test_image_method_vjp_vs_autograd)."""
    B = 257
    a = rng.normal(size=(3,)).astype(np.float32) * 3
    mvs = rng.normal(size=(k, 3)).astype(np.float32)
    mns, _ = orc.normalize(rng.normal(size=(k, 3)).astype(np.float32))
    b = (rng.normal(size=(B, 3)) * 3).astype(np.float32)
    w = torch.tensor(rng.normal(size=(B, k, 3)).astype(np.float32), device="cuda")

    def run(expand):
        ins = [torch.tensor(x, device="cuda") for x in (a, b, mvs, mns)]
        if expand:  # materialised copies: the dense path, per-element gradients
            ins = [ins[0].expand(B, 3).contiguous(), ins[1], ins[2].expand(B, k, 3).contiguous(),
                   ins[3].expand(B, k, 3).contiguous()]
        ins = [t.requires_grad_() for t in ins]
        out = G.image_method(*ins)
        (out * w).sum().backward()
        return out, [t.grad for t in ins]

    out_s, g_s = run(False)
    out_d, g_d = run(True)
    np.testing.assert_array_equal(_bits(_np(out_s)), _bits(_np(out_d)))
    np.testing.assert_array_equal(_bits(_np(g_s[1])), _bits(_np(g_d[1])))  # `to` is dense in both
    for i in (0, 2, 3):
        per = _np(g_d[i]).astype(np.float64)
        ref = per.sum(axis=0)
        assert tuple(g_s[i].shape) == ref.shape
        tol = 1e-5 * np.abs(per).sum(axis=0).max()
        np.testing.assert_allclose(_np(g_s[i]), ref, rtol=0, atol=tol)


def test_capped_survivor_queue(G):
    """drt_trace_paths_dense_capped (round 5): a survivor queue of `max_survivors` entries instead of one per row -- the
    same four arrays and counters when it fits, bit DRT_TRACE_OVERFLOW_SURVIVORS of the third counter word when it does
    not (never silent), and the true survivor count either way."""
    import ctypes as C

    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream
    from differt_amd.geometry._solvers import _params, _table_candidates

    V, Tr, tx, rx, scene = _city(G, boxes=12, ntx=3, nrx=4)
    n = scene.mesh.num_primitives
    table = torch.as_tensor(orc.generate_all_path_candidates(n, 2).astype(np.int32), device="cuda")
    ref = scene.trace_paths(path_candidates=table)
    txd, rxd = scene.transmitters.reshape(-1, 3).contiguous(), scene.receivers.reshape(-1, 3).contiguous()
    ntx, nrx, (Cn, k) = txd.shape[0], rxd.shape[0], table.shape
    params = _params(None, None, None, None)
    cands = _table_candidates(table)
    lib = _lib.load()

    def run(max_survivors):
        verts = torch.empty((ntx, nrx, Cn, k + 2, 3), dtype=torch.float32, device="cuda")
        objs = torch.empty((ntx, nrx, Cn, k + 2), dtype=torch.int32, device="cuda")
        mask = torch.empty((ntx, nrx, Cn), dtype=torch.uint8, device="cuda")
        tout = torch.empty((ntx, nrx, Cn, k), dtype=torch.int32, device="cuda")
        nbytes = lib.drt_trace_dense_capped_workspace_size(max_survivors)
        assert nbytes == 64 + 8 * max(max_survivors, 0)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        _lib.call("drt_trace_paths_dense_capped", scene.mesh.handle().h, C.byref(params), ptr(txd), ntx, ptr(rxd), nrx,
                  C.byref(cands), None, ptr(verts), ptr(objs), ptr(mask), ptr(tout), max_survivors, ptr(ws), nbytes, stream())
        torch.cuda.synchronize()
        return verts, objs, mask, ws[:24].view(torch.int64).cpu().numpy()

    survivors = None
    for cap in (ntx * nrx * Cn, 4096):
        verts, objs, mask, cnt = run(cap)
        assert torch.equal(mask.bool(), ref.mask) and torch.equal(objs, ref.objects)
        assert torch.equal(verts.view(torch.int32), ref.vertices.view(torch.int32))
        assert cnt[2] == 0 and cnt[0] - cnt[1] == int(ref.mask.sum())
        survivors = int(cnt[0])
    assert 8 < survivors <= 4096  # the small capacity above was a real bound, and the next one overflows
    verts, objs, mask, cnt = run(8)
    assert cnt[2] == _lib.DRT_TRACE_OVERFLOW_SURVIVORS and cnt[0] == survivors
    # what the 8 queue entries decided is right; rows beyond the queue were never occlusion-tested and are NOT reported
    # (mask cleared): a subset of the true mask with at most 8 rows, and the counters say how many
    assert not bool((mask.bool() & ~ref.mask).any())
    assert int(mask.sum()) == 8 - cnt[1] <= 8
