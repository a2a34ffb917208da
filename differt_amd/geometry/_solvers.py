"""Path tracers with the reference's solver interface.

Mirrors ``differt/src/differt/geometry/_solvers.py``: ``AbstractPathTracer`` (:53-247),
``ExhaustivePathTracer`` (:778-957) and the fused ``_trace_path_candidates`` (:499-770), whose
arithmetic runs in ``csrc/trace.hip`` behind ``drt_trace_paths_dense`` / ``_compact`` / ``_vjp``.
Extension for MI355X: ``trace_rank_range`` traces a window of candidate RANKS that are enumerated
on the GPU -- no candidate table is ever built -- and returns only the valid paths, compacted.
"""

from __future__ import annotations

import ctypes as C
import os
from collections.abc import Iterator, Sequence
from dataclasses import dataclass
from typing import TYPE_CHECKING, Any

import numpy as np
import torch

from .. import _lib
from .._tensors import F32_EPS, as_i32, device, ptr, stream
from ._graph import CompleteGraph, DiGraph
from ._paths import LaunchedPaths, TracedPaths
from ._utils import SizedIterator

if TYPE_CHECKING:
    from ._scene import Scene

__all__ = [
    "AbstractPathLauncher",
    "AbstractPathTracer",
    "ExhaustivePathTracer",
    "HybridPathTracer",
    "SBRPathLauncher",
]


def _params(epsilon, hit_tol, min_len, accel=None, skip_occlusion: bool = False,
            deterministic_grad: bool = False) -> _lib.TraceParams:
    if accel not in (None, "bvh"):
        raise ValueError(f"unknown accel {accel!r}")
    return _lib.TraceParams(
        10.0 * F32_EPS if epsilon is None else float(epsilon),   # _utils.py:1257-1259
        100.0 * F32_EPS if hit_tol is None else float(hit_tol),  # _utils.py:1418-1420
        10.0 * F32_EPS if min_len is None else float(min_len),   # _solvers.py:514-516
        (_lib.DRT_TRACE_USE_BVH if accel == "bvh" else 0) | (_lib.DRT_TRACE_SKIP_OCCLUSION if skip_occlusion else 0)
        | (_lib.DRT_TRACE_DETERMINISTIC_GRAD if deterministic_grad else 0),
    )


def _paths_vjp(mesh, params, tx, rx, cands, keys, gv, n: int, order: int, gtx, grx, gmv) -> None:
    """``drt_trace_paths_vjp``, or its deterministic form (stable sort + ordered sums instead of float atomics) when
    ``params`` carries ``DRT_TRACE_DETERMINISTIC_GRAD``."""
    h = mesh.handle().h
    if params is not None and (params.flags & _lib.DRT_TRACE_DETERMINISTIC_GRAD):
        nbytes = _lib.load().drt_trace_vjp_workspace_size(n, order)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=tx.device)
        _lib.call("drt_trace_paths_vjp_ex", h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx), rx.shape[0],
                  C.byref(cands), ptr(keys), ptr(gv), n, ptr(gtx), ptr(grx), ptr(gmv), ptr(ws), nbytes, stream())
    else:
        _lib.call("drt_trace_paths_vjp", h, ptr(tx), tx.shape[0], ptr(rx), rx.shape[0], C.byref(cands), ptr(keys),
                  ptr(gv), n, ptr(gtx), ptr(grx), ptr(gmv), stream())


def _table_candidates(table: torch.Tensor) -> _lib.Candidates:
    c = _lib.Candidates()
    c.table = table.data_ptr() if table.numel() else None
    c.num_candidates = table.shape[0]
    c.rank_lo, c.num_nodes, c.node_map = 0, 0, None
    c.order = table.shape[1]
    return c


def _rank_candidates(order: int, rank_lo: int, count: int, num_nodes: int,
                     node_map: torch.Tensor | None, first_map: torch.Tensor | None = None,
                     last_map: torch.Tensor | None = None, ragged: dict | None = None) -> _lib.Candidates:
    c = _lib.Candidates()
    c.table = None
    c.num_candidates = count
    c.rank_lo = rank_lo
    c.num_nodes = num_nodes
    c.node_map = None if node_map is None else node_map.data_ptr()
    c.order = order
    if first_map is not None:  # pruned product space F x N^(order-2) x L (hybrid tracer)
        # an EMPTY list still selects the product mode: keep the pointers non-NULL
        c.first_map, c.num_first = first_map.data_ptr() or 1, first_map.shape[0]
        c.last_map, c.num_last = last_map.data_ptr() or 1, last_map.shape[0]
    if ragged is not None:  # per-pair spaces in one launch: CSR offsets + prefix sums of the pair sizes
        c.pair_offsets = ragged["pair_offsets"].data_ptr()
        c.first_offsets = ragged["first_offsets"].data_ptr()
        c.last_offsets = ragged["last_offsets"].data_ptr()
        c.reserved = (1 if ragged["small"] else 0) | (2 if ragged.get("prefix") else 0)
        c.num_first = ragged["max_first"]  # grid sizing of the prefix kernel
    return c


class _TraceDenseFn(torch.autograd.Function):
    """vertices/objects/mask/interaction types for every (tx, rx, candidate) through ``drt_trace_paths_dense_ex``
    (every element is written by the kernel: outputs are allocated uninitialised); vertices differentiable in
    (tx, rx, mesh vertices) through ``drt_trace_paths_vjp``."""

    @staticmethod
    def forward(ctx, tx, rx, mesh_vertices, mesh, table, types, params):
        dev = tx.device
        ntx, nrx, (Cn, k) = tx.shape[0], rx.shape[0], table.shape
        verts = torch.empty((ntx, nrx, Cn, k + 2, 3), dtype=torch.float32, device=dev)
        objs = torch.empty((ntx, nrx, Cn, k + 2), dtype=torch.int32, device=dev)
        mask = torch.empty((ntx, nrx, Cn), dtype=torch.uint8, device=dev)
        tout = torch.empty((ntx, nrx, Cn, k), dtype=torch.int32, device=dev)
        counts = torch.zeros(2, dtype=torch.int64, device=dev)
        if ntx * nrx * Cn:
            lib = _lib.load()
            nbytes = lib.drt_trace_dense_workspace_size(ntx, nrx, Cn)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            cands = _table_candidates(table)
            _lib.call("drt_trace_paths_dense_ex", mesh.handle().h, C.byref(params), ptr(tx), ntx, ptr(rx),
                      nrx, C.byref(cands), ptr(types), ptr(verts), ptr(objs), ptr(mask), ptr(tout), ptr(ws), nbytes,
                      stream())
            # device-side counters of the call (include/differt_amd.h): survivors of the geometric checks, and those of
            # them the occlusion stage cleared -- TracedPaths.num_valid_paths without a pass over the mask.  A fifth,
            # non-differentiable OUTPUT of this function (a 16-byte copy: the 8 B / row workspace is not kept alive),
            # not a side channel on the mesh object: two tracers that share a mesh cannot swap counts.
            counts.copy_(ws[:16].view(torch.int64))
        ctx.mesh, ctx.table, ctx.params = mesh, table, params
        ctx.save_for_backward(tx, rx)
        ctx.mark_non_differentiable(objs, mask, tout, counts)
        return verts, objs, mask, tout, counts

    @staticmethod
    def backward(ctx, gv, _go, _gm, _gt, _gc):
        tx, rx = ctx.saved_tensors
        mesh, table = ctx.mesh, ctx.table
        n = gv.numel() // (3 * (table.shape[1] + 2))
        gtx, grx = torch.zeros_like(tx), torch.zeros_like(rx)
        gmv = torch.zeros_like(mesh.vertices) if ctx.needs_input_grad[2] else None
        if n:
            keys = torch.arange(n, dtype=torch.int64, device=tx.device)
            _paths_vjp(mesh, ctx.params, tx, rx, _table_candidates(table), keys, gv.contiguous(), n, table.shape[1],
                       gtx, grx, gmv)
        return gtx, grx, gmv, None, None, None, None


class _TraceCompactFn(torch.autograd.Function):
    """Valid paths only (sorted like ``masked_vertices``); differentiable like the dense form."""

    @staticmethod
    def forward(ctx, tx, rx, mesh_vertices, mesh, cand_desc, params, max_survivors, max_paths):
        dev = tx.device
        order = cand_desc["order"]
        lib = _lib.load()

        def make_cands():
            if cand_desc["table"] is not None:
                c = _table_candidates(cand_desc["table"])
                if cand_desc.get("pair_offsets") is not None:  # per-pair table (beam-pruned rows)
                    c.pair_offsets = cand_desc["pair_offsets"].data_ptr()
                return c
            return _rank_candidates(order, cand_desc["rank_lo"], cand_desc["count"],
                                    cand_desc["num_nodes"], cand_desc["node_map"],
                                    cand_desc.get("first_map"), cand_desc.get("last_map"),
                                    cand_desc.get("ragged"))

        while True:
            nbytes = lib.drt_trace_compact_workspace_size(max_survivors, max_paths)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            keys = torch.empty(max_paths, dtype=torch.int64, device=dev)
            verts = torch.empty((max_paths, order + 2, 3), dtype=torch.float32, device=dev)
            objs = torch.empty((max_paths, order + 2), dtype=torch.int32, device=dev)
            nv = C.c_int64(0)
            cands = make_cands()
            try:
                _lib.call("drt_trace_paths_compact", mesh.handle().h, C.byref(params), ptr(tx),
                          tx.shape[0], ptr(rx), rx.shape[0], C.byref(cands), max_survivors, max_paths,
                          ptr(keys), ptr(verts), ptr(objs), C.byref(nv), ptr(ws), nbytes, stream())
                break
            except _lib.CapacityError:
                # the call reports what it needs; grow and retry (results never depend on capacities)
                need = int(nv.value)
                if need > max_survivors:
                    max_survivors = max(2 * max_survivors, need)
                else:
                    max_paths = max(2 * max_paths, need)
        n = int(nv.value)
        ctx.mesh, ctx.make_cands, ctx.params, ctx.order = mesh, make_cands, params, order
        keys = keys[:n].clone()
        ctx.save_for_backward(tx, rx, keys)
        objs = objs[:n].clone()
        ctx.mark_non_differentiable(objs, keys)
        return verts[:n].clone(), objs, keys

    @staticmethod
    def backward(ctx, gv, _go, _gk):
        tx, rx, keys = ctx.saved_tensors
        mesh = ctx.mesh
        gtx, grx = torch.zeros_like(tx), torch.zeros_like(rx)
        gmv = torch.zeros_like(mesh.vertices) if ctx.needs_input_grad[2] else None
        n = keys.shape[0]
        if n:
            _paths_vjp(mesh, ctx.params, tx, rx, ctx.make_cands(), keys, gv.contiguous(), n, ctx.order, gtx, grx, gmv)
        return gtx, grx, gmv, None, None, None, None, None


class _TraceBeamFn(torch.autograd.Function):
    """``drt_trace_paths_beam``: the valid paths of the exhaustive tracer through the geometric pruning, ONE
    native call; differentiable through ``drt_trace_paths_vjp`` on the returned (self-describing) keys."""

    @staticmethod
    def forward(ctx, tx, rx, mesh_vertices, mesh, order, params, beam, max_paths, workspace):
        dev = tx.device
        lib = _lib.load()
        n = mesh.num_primitives
        h = mesh.handle().h
        while True:
            nbytes = lib.drt_trace_beam_workspace_size(tx.shape[0], rx.shape[0], n, order, C.byref(beam), max_paths)
            ws = workspace(nbytes, dev)
            keys = torch.empty(max_paths, dtype=torch.int64, device=dev)
            verts = torch.empty((max_paths, order + 2, 3), dtype=torch.float32, device=dev)
            objs = torch.empty((max_paths, order + 2), dtype=torch.int32, device=dev)
            nv = C.c_int64(0)
            try:
                _lib.call("drt_trace_paths_beam", h, C.byref(params), C.byref(beam), ptr(tx), tx.shape[0], ptr(rx),
                          rx.shape[0], order, max_paths, ptr(keys), ptr(verts), ptr(objs), C.byref(nv), ptr(ws),
                          nbytes, stream())
                break
            except _lib.CapacityError as exc:
                # only the output capacity is grown here; list capacities are the caller's (results never
                # depend on capacities)
                if "max_paths" not in exc.msg:
                    raise
                max_paths = max(2 * max_paths, int(nv.value))
        nvalid = int(nv.value)
        keys = keys[:nvalid].clone()
        objs = objs[:nvalid].clone()
        ctx.mesh, ctx.order, ctx.n, ctx.params = mesh, order, n, params
        ctx.save_for_backward(tx, rx, keys)
        ctx.mark_non_differentiable(objs, keys)
        return verts[:nvalid].clone(), objs, keys

    @staticmethod
    def backward(ctx, gv, _go, _gk):
        tx, rx, keys = ctx.saved_tensors
        mesh = ctx.mesh
        gtx, grx = torch.zeros_like(tx), torch.zeros_like(rx)
        gmv = torch.zeros_like(mesh.vertices) if ctx.needs_input_grad[2] else None
        if keys.shape[0]:
            if ctx.order == 0:
                cands = _rank_candidates(0, 0, 1, max(ctx.n, 1), None)
            else:
                cands = _lib.Candidates()
                cands.table, cands.num_nodes, cands.order = None, ctx.n, ctx.order
                cands.reserved = _lib.DRT_CAND_PACKED_KEYS
            _paths_vjp(mesh, ctx.params, tx, rx, cands, keys, gv.contiguous(), keys.shape[0], ctx.order, gtx, grx, gmv)
        return gtx, grx, gmv, None, None, None, None, None, None


class _TraceHybridPairsFn(torch.autograd.Function):
    """``drt_trace_paths_hybrid_pairs``: per-pair visibility-pruned trace (CSR sets and pair offsets built by kernels in
    the call's workspace); keys are packed, the backward is ``drt_trace_paths_vjp`` with ``DRT_CAND_PACKED_KEYS``."""

    @staticmethod
    def forward(ctx, tx, rx, mesh_vertices, mesh, order, params, vis_tx, vis_rx, flags, max_survivors, max_paths, info):
        dev = tx.device
        lib = _lib.load()
        n = mesh.num_primitives
        h = mesh.handle().h
        while True:
            nbytes = lib.drt_trace_hybrid_pairs_workspace_size(tx.shape[0], rx.shape[0], n, max_survivors, max_paths)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            keys = torch.empty(max_paths, dtype=torch.int64, device=dev)
            verts = torch.empty((max_paths, order + 2, 3), dtype=torch.float32, device=dev)
            objs = torch.empty((max_paths, order + 2), dtype=torch.int32, device=dev)
            nv, ne = C.c_int64(0), C.c_int64(0)
            try:
                _lib.call("drt_trace_paths_hybrid_pairs", h, C.byref(params), ptr(tx), tx.shape[0], ptr(rx), rx.shape[0],
                          order, ptr(vis_tx), ptr(vis_rx), flags, max_survivors, max_paths, ptr(keys), ptr(verts),
                          ptr(objs), C.byref(nv), C.byref(ne), ptr(ws), nbytes, stream())
                break
            except _lib.CapacityError:  # the call reports the count that did not fit (results never depend on capacities)
                need = int(nv.value)
                if need > max_survivors:
                    max_survivors = max(2 * max_survivors, need)
                else:
                    max_paths = max(2 * max_paths, need)
        info["evaluated"] = int(ne.value)
        nvalid = int(nv.value)
        keys = keys[:nvalid].clone()
        objs = objs[:nvalid].clone()
        ctx.mesh, ctx.order, ctx.n, ctx.params = mesh, order, n, params
        ctx.save_for_backward(tx, rx, keys)
        ctx.mark_non_differentiable(objs, keys)
        return verts[:nvalid].clone(), objs, keys

    @staticmethod
    def backward(ctx, gv, _go, _gk):
        tx, rx, keys = ctx.saved_tensors
        mesh = ctx.mesh
        gtx, grx = torch.zeros_like(tx), torch.zeros_like(rx)
        gmv = torch.zeros_like(mesh.vertices) if ctx.needs_input_grad[2] else None
        if keys.shape[0]:
            cands = _lib.Candidates()
            cands.table, cands.num_nodes, cands.order = None, ctx.n, ctx.order
            cands.reserved = _lib.DRT_CAND_PACKED_KEYS
            _paths_vjp(mesh, ctx.params, tx, rx, cands, keys, gv.contiguous(), keys.shape[0], ctx.order, gtx, grx, gmv)
        return gtx, grx, gmv, None, None, None, None, None, None, None, None, None


class _TraceSmoothFn(torch.autograd.Function):
    """Smoothed tracer (_solvers.py:499-770 with ``smoothing_factor``): vertices AND the float mask are
    differentiable in (tx, rx, mesh vertices) through ``drt_trace_paths_dense_smooth_vjp``."""

    @staticmethod
    def forward(ctx, tx, rx, mesh_vertices, mesh, table, params, alpha, batch_size):
        dev = tx.device
        ntx, nrx, (Cn, k) = tx.shape[0], rx.shape[0], table.shape
        verts = torch.zeros((ntx, nrx, Cn, k + 2, 3), dtype=torch.float32, device=dev)
        objs = torch.zeros((ntx, nrx, Cn, k + 2), dtype=torch.int32, device=dev)
        mask = torch.zeros((ntx, nrx, Cn), dtype=torch.float32, device=dev)
        if ntx * nrx * Cn:
            cands = _table_candidates(table)
            _lib.call("drt_trace_paths_dense_smooth", mesh.handle().h, C.byref(params), alpha, batch_size,
                      ptr(tx), ntx, ptr(rx), nrx, C.byref(cands), ptr(verts), ptr(objs), ptr(mask), stream())
        ctx.mesh, ctx.table, ctx.cfg = mesh, table, (params, alpha, batch_size)
        ctx.save_for_backward(tx, rx)
        ctx.mark_non_differentiable(objs)
        return verts, objs, mask

    @staticmethod
    def backward(ctx, gv, _go, gm):
        tx, rx = ctx.saved_tensors
        mesh, table = ctx.mesh, ctx.table
        params, alpha, batch_size = ctx.cfg
        gtx, grx = torch.zeros_like(tx), torch.zeros_like(rx)
        gmv = torch.zeros_like(mesh.vertices) if ctx.needs_input_grad[2] else None
        if gv.numel() // (3 * (table.shape[1] + 2)):
            cands = _table_candidates(table)
            _lib.call("drt_trace_paths_dense_smooth_vjp", mesh.handle().h, C.byref(params), alpha, batch_size,
                      ptr(tx), tx.shape[0], ptr(rx), rx.shape[0], C.byref(cands), ptr(gv.contiguous()),
                      ptr(gm.contiguous()), ptr(gtx), ptr(grx), ptr(gmv), stream())
        return gtx, grx, gmv, None, None, None, None, None


def _trace_path_candidates(mesh, tx_vertices, rx_vertices, path_candidates, interaction_types=None, *,
                           epsilon, hit_tol, min_len, smoothing_factor, confidence_threshold,
                           batch_size, accel=None, deterministic_grad: bool = False, stats=None) -> TracedPaths:
    """Reference ``_trace_path_candidates`` (_solvers.py:499-770), dense layout
    ``[num_tx, num_rx, num_candidates, ...]``; ``smoothing_factor`` switches to the float-mask mode
    (:599-713), whose blocked term uses the pure operator with ``batch_size`` tiles (:665-674)."""
    table = as_i32(path_candidates).contiguous()
    tx = tx_vertices.contiguous()
    rx = rx_vertices.contiguous()
    if interaction_types is not None:
        types = as_i32(interaction_types)
        if tuple(types.shape) != tuple(table.shape):  # the reference broadcasts [C, order] over (tx, rx), SV:751-757
            types = types.expand(table.shape)
        types = types.contiguous()
    else:
        types = None  # all specular reflections (0), SV:758-762
    if smoothing_factor is not None:
        verts, objs, mask = _TraceSmoothFn.apply(
            tx, rx, mesh.vertices, mesh, table, _params(epsilon, hit_tol, min_len, None),
            float(smoothing_factor), 0 if batch_size is None else int(batch_size))
        if types is None:
            it = torch.zeros(objs.shape[:-1] + (table.shape[1],), dtype=torch.int32, device=objs.device)
        else:
            it = types.expand(*objs.shape[:-1], table.shape[1])
        return TracedPaths(verts, objs, mask, it, confidence_threshold)
    params = _params(epsilon, hit_tol, min_len, accel, deterministic_grad=deterministic_grad)
    if stats is not None:  # drt_trace_stats of the dense call: HIP-event kernel times, one stream synchronisation
        params.stats = C.pointer(stats)
    verts, objs, mask, it, counts = _TraceDenseFn.apply(tx, rx, mesh.vertices, mesh, table, types, params)
    # the kernel writes 0 / 1 bytes: reinterpret, do not copy
    paths = TracedPaths(verts, objs, mask.view(torch.bool), it, confidence_threshold)
    if mask.numel():
        paths._attach_valid_count(counts)
    return paths


def Scene_like(scene, tx, rx):
    """``scene`` with other end points (same mesh object)."""
    return type(scene)(tx, rx, scene.mesh)


class AbstractPathTracer:
    """Solver interface of the reference (_solvers.py:53-247): any object with these methods can be
    passed as ``Scene.trace_paths(solver=...)``."""

    epsilon: float | None = None
    hit_tol: float | None = None

    def generate_path_candidates(self, scene, order, specular_reflection=True, diffuse_scattering=False):
        raise NotImplementedError

    def trace_path_candidates(self, scene, path_candidates, interaction_types):
        raise NotImplementedError

    def generate_path_candidates_chunks_iter(self, scene, order, *args, chunk_size, pad_chunks=False,
                                             **kwargs):
        """Default of the reference (_solvers.py:93-174): slice the full table."""
        cands, types = self.generate_path_candidates(scene, order, *args, **kwargs)
        n = cands.shape[0]
        nchunks = -(-n // chunk_size) if n else 0

        def gen() -> Iterator:
            for i in range(nchunks):
                c, t = cands[i * chunk_size:(i + 1) * chunk_size], types[i * chunk_size:(i + 1) * chunk_size]
                if pad_chunks and c.shape[0] < chunk_size:
                    pad = chunk_size - c.shape[0]
                    c = torch.cat((c, torch.full((pad, c.shape[1]), -1, dtype=c.dtype, device=c.device)))
                    t = torch.cat((t, torch.zeros((pad, t.shape[1]), dtype=t.dtype, device=t.device)))
                yield c, t

        return SizedIterator(gen(), size=nchunks)

    def trace_paths(self, scene, order, chunk_size: int | None = None, pad_chunks: bool = False):
        """Generate then trace (_solvers.py:214-247): one :class:`TracedPaths`, or an iterator of
        them (one per chunk) when ``chunk_size`` is given.  Layout ``[num_tx, num_rx, C, ...]``."""
        if chunk_size is not None:
            return (self.trace_path_candidates(scene, c, t)
                    for c, t in self.generate_path_candidates_chunks_iter(
                        scene, order, chunk_size=chunk_size, pad_chunks=pad_chunks))
        cands, types = self.generate_path_candidates(scene, order)
        return self.trace_path_candidates(scene, cands, types)


@dataclass
class ExhaustivePathTracer(AbstractPathTracer):
    """Exhaustive image-method tracer (reference _solvers.py:778-957), same fields and defaults."""

    epsilon: float | None = None
    hit_tol: float | None = None
    min_len: float | None = None
    smoothing_factor: float | None = None
    confidence_threshold: float = 0.5
    batch_size: int | None = 512
    disconnect_inactive_triangles: bool = False
    chunk_size: int | None = None
    accel: str | None = None
    """MI355X extension: ``"bvh"`` makes the occlusion stage walk the mesh LBVH (O(log T) per segment,
    like the reference's Warp path) instead of testing every triangle."""
    deterministic_grad: bool = False
    """MI355X extension (SURVEY.md section 7, hard part 6): gradients of the traced vertices are summed in a fixed
    order (``DRT_TRACE_DETERMINISTIC_GRAD``: stable sort by destination + ordered sums) instead of float atomics --
    bit-identical from run to run, a few kernel launches slower."""
    collect_stats: bool = False
    """Fill :attr:`last_stats` (``drt_trace_stats``: candidates / survivors / valid paths and the
    HIP-event time of the filter, occlusion and sort+emit stages) on every compact trace; costs two
    extra stream synchronisations per call."""
    literal: bool = False
    """MI355X extension (round 6).  ``False``: a compact trace of a WHOLE candidate space of order 1..3
    (``Scene.trace_paths(order, compact=True)``, ``trace_rank_range(scene, order)``) runs through the geometrically pruned
    search (``drt_trace_paths_beam``: the same paths, order, vertex bits and keys, 100-1000x faster at 10k triangles;
    DESIGN.md section 9 -- its one documented exclusion are float artifacts with two reflection points closer than
    64 ulp(M), section 9.8).  ``True``: every candidate is evaluated by the filter kernel, literally as the reference
    enumerates them (_solvers.py:803-848, 936-957)."""

    # ---- candidate generation (host graph classes; lexicographic like graph.rs) ----
    def _graph(self, scene):
        mesh = scene.mesh
        graph = CompleteGraph(mesh.num_primitives)
        if self.disconnect_inactive_triangles and mesh.mask is not None:  # _solvers.py:820-827
            mask = mesh.mask
            if mesh.assume_quads:
                mask = mask[0::2] & mask[1::2]
            graph = DiGraph.from_complete_graph(graph)
            from_, to = graph.insert_from_and_to_nodes()
            graph.filter_by_mask(mask.cpu().numpy(), fast_mode=True)
        else:
            from_, to = graph.num_nodes, graph.num_nodes + 1
        return graph, from_, to

    def generate_path_candidates(self, scene, order, specular_reflection=True,  # noqa: ARG002
                                 diffuse_scattering=False):  # noqa: ARG002
        """``(candidates i32[C, order], interaction_types i32[C, order])`` (_solvers.py:803-848)."""
        if isinstance(order, Sequence):
            raise NotImplementedError("ExhaustivePathTracer does not support multiple orders yet.")
        if order == 0:
            # the one direct path from -> to, whatever the graph between them looks like (complete, masked or visibility-
            # pruned: insert_from_and_to_nodes(direct_path=True), graph.rs) -- no 7 000 x 7 000 adjacency matrix for it
            # (the reference's harness traces orders 0 and 1 on bruxelles.obj, tests/benchmarks/test_rt.py:151-196)
            cands = torch.empty((1, 0), dtype=torch.int32, device=device())
            return cands, torch.zeros_like(cands)
        fast = self._fast_candidates(scene, order)
        if fast is not None:
            return fast, torch.zeros_like(fast)
        if type(self) is ExhaustivePathTracer and order >= 1:
            # complete graph (optionally over the active primitives only): the table is unranked on
            # the GPU, no host enumeration and no host->device copy (reference: _solvers.py:817-843)
            n, node_map = self._num_nodes_and_map(scene)
            total = n * (n - 1) ** (order - 1) if n > 0 else 0
            if total >= 2**31:
                raise MemoryError(f"{total} candidates: use trace_rank_range / chunk_size instead of a table")
            cands = torch.empty((total, order), dtype=torch.int32, device=device())
            if total:
                _lib.call("drt_candidates_fill", max(n, 1), order, 0, total, ptr(node_map),
                          2 if scene.mesh.assume_quads else 1, ptr(cands), stream())
            return cands, torch.zeros_like(cands)
        graph, from_, to = self._graph(scene)
        arr = graph.all_paths_array(from_, to, order + 2, include_from_and_to=False)
        cands = torch.as_tensor(arr.astype(np.int32).reshape(arr.shape[0], order), device=device())
        if scene.mesh.assume_quads:
            cands = 2 * cands  # _solvers.py:842-843
        return cands, torch.zeros_like(cands)

    def _fast_candidates(self, scene, order):  # noqa: ARG002 - subclasses with a pruned graph override
        return None

    def generate_path_candidates_chunks_iter(self, scene, order, *args, chunk_size=None,
                                             pad_chunks=False, **kwargs):
        """Native chunked generation (_solvers.py:850-934); the last chunk is padded with ``-1`` rows
        when ``pad_chunks`` (:912-918)."""
        eff = chunk_size or self.chunk_size
        if eff is None:
            return SizedIterator(iter([self.generate_path_candidates(scene, order, *args, **kwargs)]), size=1)
        if isinstance(order, Sequence):
            raise NotImplementedError("ExhaustivePathTracer does not support multiple orders yet.")
        quads = scene.mesh.assume_quads
        if type(self) is ExhaustivePathTracer and order >= 1:
            # complete graph (optionally over the active primitives only): every chunk is a rank interval unranked
            # on the GPU straight into the chunk's table -- no host enumeration, no host->device copy
            # (reference: Rust iterator + transfer, _solvers.py:870-934); same rows, same order (graph.rs:400-470)
            n, node_map = self._num_nodes_and_map(scene)
            total = n * (n - 1) ** (order - 1) if n > 0 else 0
            if total > 0:
                eff = int(eff)
                nchunks = -(-total // eff)

                def gen_gpu() -> Iterator:
                    dev = device()
                    for i in range(nchunks):
                        lo, hi = i * eff, min((i + 1) * eff, total)
                        if pad_chunks and hi - lo < eff:  # _solvers.py:912-918
                            # (the reference pads with -1 and THEN doubles quad ids, :912-925: padding rows of a quad mesh are -2)
                            c = torch.full((eff, order), -2 if quads else -1, dtype=torch.int32, device=dev)
                        else:
                            c = torch.empty((hi - lo, order), dtype=torch.int32, device=dev)
                        _lib.call("drt_candidates_fill", n, order, lo, hi, ptr(node_map), 2 if quads else 1, ptr(c),
                                  stream())
                        yield c, torch.zeros_like(c)

                return SizedIterator(gen_gpu(), size=nchunks)
        graph, from_, to = self._graph(scene)
        it = graph.all_paths_array_chunks(from_, to, order + 2, include_from_and_to=False, chunk_size=eff)

        def gen() -> Iterator:
            for chunk in it:
                arr = np.asarray(chunk).astype(np.int32)
                arr = arr.reshape(arr.shape[0], order)
                if pad_chunks and arr.shape[0] < eff:
                    arr = np.pad(arr, ((0, eff - arr.shape[0]), (0, 0)), constant_values=-1)
                c = torch.as_tensor(arr, device=device())
                if quads:
                    c = 2 * c
                yield c, torch.zeros_like(c)

        size: Any = it.__len__ if hasattr(it, "__len__") else -1
        return SizedIterator(gen(), size=size)

    # ---- tracing ----
    def trace_path_candidates(self, scene, path_candidates, interaction_types=None) -> TracedPaths:
        """_solvers.py:936-957."""
        st = _lib.TraceStats() if (self.collect_stats and self.smoothing_factor is None) else None
        out = _trace_path_candidates(
            scene.mesh, scene.transmitters.reshape(-1, 3), scene.receivers.reshape(-1, 3),
            path_candidates, interaction_types, epsilon=self.epsilon, hit_tol=self.hit_tol,
            min_len=self.min_len, smoothing_factor=self.smoothing_factor,
            confidence_threshold=self.confidence_threshold, batch_size=self.batch_size,
            accel=self.accel, deterministic_grad=self.deterministic_grad, stats=st,
        )
        if st is not None:
            self.last_stats = {f: getattr(st, f) for f, _ in _lib.TraceStats._fields_ if f != "reserved"}
        return out

    def num_path_candidates(self, scene, order: int) -> int:
        """``n * (n-1)**(order-1)`` over the (active) primitives; 1 for order 0."""
        n = self._num_nodes_and_map(scene)[0]
        return 1 if order == 0 else n * (n - 1) ** (order - 1)

    def _num_nodes_and_map(self, scene):
        mesh = scene.mesh
        if self.disconnect_inactive_triangles and mesh.mask is not None:
            mask = mesh.mask
            if mesh.assume_quads:
                mask = mask[0::2] & mask[1::2]
            node_map = torch.nonzero(mask).reshape(-1).to(torch.int32).contiguous()
            return int(node_map.shape[0]), node_map
        return mesh.num_primitives, None

    def trace_rank_range(self, scene, order: int, rank_lo: int = 0, rank_hi: int | None = None, *,
                         max_survivors: int = 1 << 20, max_paths: int = 1 << 16, literal: bool | None = None) -> TracedPaths:
        """Trace candidates ``[rank_lo, rank_hi)`` of the lexicographic candidate order without
        materialising them; returns the valid paths only, in ``masked_vertices`` order.
        ``keys`` holds ``(tx*num_rx + rx) * (rank_hi - rank_lo) + (rank - rank_lo)``.

        The WHOLE space of an order 1..3 goes through the pruned search unless ``literal`` (default: the tracer's
        :attr:`literal`) is true -- same paths, same keys (see :attr:`literal`)."""
        if self.smoothing_factor is not None:
            raise NotImplementedError("the smoothed mode is dense by nature (every candidate gets a confidence): "
                                      "use trace_path_candidates / Scene.trace_paths without compact")
        n, node_map = self._num_nodes_and_map(scene)
        total = 1 if order == 0 else n * (n - 1) ** (order - 1)
        hi = total if rank_hi is None else min(int(rank_hi), total)
        lo = min(int(rank_lo), hi)
        lit = self.literal if literal is None else bool(literal)
        if (not lit and type(self) is ExhaustivePathTracer and 1 <= order <= 3 and lo == 0 and hi == total and total > 0
                and scene.transmitters.numel() and scene.receivers.numel()
                and scene.transmitters.reshape(-1, 3).shape[0] * scene.receivers.reshape(-1, 3).shape[0]
                * max(scene.mesh.num_primitives, 1) ** order < 2 ** 62):
            return self._rank_keyed(scene, self.trace_beam_pruned(scene, order, max_paths=max_paths), order, n, node_map, total)
        desc = {"table": None, "order": order, "rank_lo": lo, "count": hi - lo, "num_nodes": max(n, 1),
                "node_map": node_map}
        return self._trace_compact(scene, desc, max_survivors, max_paths)

    def trace_rank_range_literal(self, *args, **kwargs) -> TracedPaths:
        """:meth:`trace_rank_range` with every candidate evaluated (``literal=True``): what the tests, the stress drivers
        and the exhaustive legs of the benches use as the EXHAUSTIVE side of a comparison."""
        return self.trace_rank_range(*args, literal=True, **kwargs)

    def _rank_keyed(self, scene, p: TracedPaths, order: int, n: int, node_map, total: int) -> TracedPaths:
        """Paths of the pruned search with the keys of :meth:`trace_rank_range`: ``(tx*num_rx + rx) * total + rank``,
        ``rank`` = position of ``(m_1 .. m_k)`` in the lexicographic enumeration of the complete graph over the nodes
        (graph.rs:301-397): ``m_1 (n-1)^(k-1) + sum_j (m_j - [m_j > m_(j-1)]) (n-1)^(k-j)``.  Both key orders are the
        lexicographic order of ``(tx, rx, m_1 .. m_k)``, so the rows keep their places."""
        objs = p.objects
        ids = objs[:, 1:-1].to(torch.int64)
        mesh = scene.mesh
        if mesh.assume_quads:
            ids = ids // 2  # objects report the even triangle id of a quad (_solvers.py:736-744)
        if node_map is not None:  # disconnect_inactive_triangles: nodes = active primitives, in order
            inv = torch.full((max(mesh.num_primitives, 1),), -1, dtype=torch.int64, device=ids.device)
            inv[node_map.to(torch.int64)] = torch.arange(node_map.shape[0], dtype=torch.int64, device=ids.device)
            ids = inv[ids]
        rank = ids[:, 0].clone() if order >= 1 else torch.zeros(objs.shape[0], dtype=torch.int64, device=objs.device)
        for j in range(1, order):
            rank = rank * (n - 1) + (ids[:, j] - (ids[:, j] > ids[:, j - 1]).to(torch.int64))
        nrx = scene.receivers.reshape(-1, 3).shape[0]
        keys = (objs[:, 0].to(torch.int64) * nrx + objs[:, -1].to(torch.int64)) * total + rank
        return TracedPaths(p.vertices, p.objects, p.mask, p.interaction_types, p.confidence_threshold, keys)

    def trace_path_candidates_compact(self, scene, path_candidates, *, max_survivors: int = 1 << 20,
                                      max_paths: int = 1 << 16) -> TracedPaths:
        """Compacted variant of :meth:`trace_path_candidates` for an explicit table."""
        table = as_i32(path_candidates).contiguous()
        desc = {"table": table, "order": table.shape[1]}
        return self._trace_compact(scene, desc, max_survivors, max_paths)

    # ---- conservative ("beam") pruning: the lossless counterpart of the hybrid tracer's sampling ----
    def trace_beam_pruned(self, scene, order: int, *, kappa: float = 64.0, expansion: str = "auto", emit: str = "auto",
                          max_entries: int | None = None, max_records: int | None = None, max_rows: int | None = None,
                          max_survivors: int | None = None, max_paths: int = 1 << 16, probe_prefixes: int | None = None,
                          prefix_shard: tuple[int, int] | None = None, pairs: bool = True,
                          rows: str = "auto") -> TracedPaths:
        """The valid paths of the exhaustive tracer -- same objects, same ``masked_vertices`` order, identical
        vertex bits, same autograd -- without visiting ``n (n-1)**(order-1)`` candidates per pair: ONE call of
        ``drt_trace_paths_beam`` (csrc/beam.hip; reference context: the exhaustive enumeration
        _solvers.py:803-848 traced by :936-957, and the SAMPLED pruning of the hybrid tracer :1013-1056).

        A prefix of mirrors is dropped only when a necessary condition of a valid specular path -- next
        primitive inside the pyramids spanned by the image of the transmitter and the mirrors so far; previous
        and next point on one side of the mirror plane (the reference's same-side check) -- fails by more than a
        bound on the error of the reference's own float32 reflection points, built per mirror as
        ``kappa * ulp(M) * sigma * D / h`` from its incidence geometry: a mirror seen at
        grazing incidence switches its own tests off, there is no smallest incidence angle to choose.
        ``keys`` are ``(tx*num_rx + rx) * n**order + sum_j m_j * n**(order-1-j)`` (``n`` primitives, ``m_j``
        primitive ids).  Orders 0..3.

        ``expansion``: ``"clustered"`` (= ``"auto"``: primitives in Morton clusters of 64, box test per (prefix,
        cluster); the last expansion of orders 2 and 3 runs as two kernels, ``"fused"`` keeps it in one) or ``"plain"`` (every
        pair tested); ``emit``: ``"plain"``, ``"clustered"`` (receivers in Morton
        clusters) or ``"auto"`` (clustered from 128 receivers on) -- the same rows either way.
        ``prefix_shard=(rank, world)`` keeps the level-1 prefixes (transmitter ``t``, first mirror ``m``) with
        ``(t * n + m) % world == rank``: the multi-GPU split of
        ``differt_amd.distributed.trace_beam_pruned_sharded`` -- every valid path has exactly one level-1
        prefix, so the shards' results partition the full result; rank 0 owns the line-of-sight paths.
        ``pairs=False`` (``DRT_BEAM_NO_PAIRS``) searches a triangle mesh triangle by triangle even when the pairing pass
        found coplanar pairs (two triangles ``(v0, v1, v2)``, ``(v0, v2, v3)`` anywhere in the mesh with equal unit
        normals, first vertices and mask values and a convex union: the same mirror bit for bit -- the walls of a box
        city, the walls and consecutive roof ears of the reference's bruxelles.obj), which the search otherwise runs
        over as single primitives -- the same result either way (tested), a quarter of the level-2 prefixes on a box
        city.  In that mode the exact
        trace evaluates the image chain once per surviving pair row and tests both triangles of every pair
        (``DRT_CAND_PAIR_BLOCKS``); ``rows="plain"`` (``DRT_BEAM_ROWS_PLAIN``) traces the ``2**order`` triangle rows one
        by one instead -- the same result (tested)."""
        if self.smoothing_factor is not None:
            raise NotImplementedError("the smoothed mode is dense by nature: use trace_path_candidates")
        if not 0 <= order <= 3:
            raise ValueError("beam pruning covers orders 0..3")
        if expansion not in ("auto", "clustered", "plain", "fused"):
            raise ValueError(f"unknown expansion {expansion!r}")
        if emit not in ("auto", "plain", "clustered"):
            raise ValueError(f"unknown emit {emit!r}")
        if rows not in ("auto", "plain"):
            raise ValueError(f"unknown rows {rows!r}")
        beam = _lib.BeamParams()
        beam.kappa = float(kappa)
        beam.flags = ((_lib.DRT_BEAM_EXPAND_PLAIN if expansion == "plain" else 0)
                      | (_lib.DRT_BEAM_EXPAND_FUSED if expansion == "fused" else 0)
                      | (_lib.DRT_BEAM_EMIT_PLAIN if emit == "plain" else 0)
                      | (_lib.DRT_BEAM_EMIT_CLUSTERED if emit == "clustered" else 0)
                      | (0 if pairs else _lib.DRT_BEAM_NO_PAIRS)
                      | (_lib.DRT_BEAM_ROWS_PLAIN if rows == "plain" else 0))
        beam.max_entries, beam.max_records = int(max_entries or 0), int(max_records or 0)
        beam.max_rows, beam.max_survivors = int(max_rows or 0), int(max_survivors or 0)
        # slice size of the last expansion: the caller's, else what the previous call on this (mesh, order, end-point
        # counts) had settled on (drt_beam_stats.next_probe_prefixes) -- a loop that moves its transmitters starts with
        # full-size slices instead of a 4096-prefix probe; only a hint (an overflowing slice is retried smaller)
        hint_key = (scene.mesh.generation(), order, bool(pairs), int(scene.transmitters.reshape(-1, 3).shape[0]),
                    int(scene.receivers.reshape(-1, 3).shape[0]), float(kappa), prefix_shard)
        hints = self.__dict__.setdefault("_beam_probe_hints", {})
        beam.probe_prefixes = int(probe_prefixes or (0 if os.environ.get("DRT_BEAM_NO_HINT") else hints.get(hint_key, 0)))
        if prefix_shard is not None:
            srank, sworld = int(prefix_shard[0]), int(prefix_shard[1])
            if sworld <= 0 or not 0 <= srank < sworld:
                raise ValueError("prefix_shard = (rank, world) with 0 <= rank < world")
            beam.shard_rank, beam.shard_world = srank, sworld
        st = _lib.BeamStats()
        beam.stats = C.pointer(st)
        tx = scene.transmitters.reshape(-1, 3).contiguous()
        rx = scene.receivers.reshape(-1, 3).contiguous()
        mesh = scene.mesh
        if tx.shape[0] * rx.shape[0] * max(mesh.num_primitives, 1) ** order >= 2 ** 62:
            raise OverflowError("tx * rx * primitives**order does not fit a 62-bit row key")
        verts, objs, keys = _TraceBeamFn.apply(tx, rx, mesh.vertices, mesh, order, _params(self.epsilon, self.hit_tol, self.min_len, self.accel, deterministic_grad=self.deterministic_grad),
                                               beam, int(max_paths), self._beam_workspace)
        self.last_beam_stats = {"unit_m": st.unit_m, "magnitude": st.magnitude, "levels": [int(x) for x in st.levels[:max(order, 1)]],
                                "rows": int(st.rows), "chunks": int(st.slices), "valid": int(st.valid),
                                "grazing_prefixes": int(st.grazing_prefixes), "pair_mode": bool(st.pair_mode),
                                "paired_primitives": int(st.paired_primitives),
                                "expand_last_ms": float(st.expand_last_ms), "emit_ms": float(st.emit_ms),
                                "trace_ms": float(st.trace_ms)}
        if order >= 2 and st.next_probe_prefixes >= 1.0:
            if len(hints) > 64:
                hints.clear()
            hints[hint_key] = int(st.next_probe_prefixes)
        nv = objs.shape[0]
        return TracedPaths(verts, objs, torch.ones(nv, dtype=torch.bool, device=objs.device),
                           torch.zeros((nv, order), dtype=torch.int32, device=objs.device), self.confidence_threshold, keys)

    def trace_beam_pruned_static(self, scene, order: int, *, max_paths: int, kappa: float = 64.0,
                                 max_entries: int | None = None, max_records: int | None = None,
                                 max_rows: int | None = None, max_survivors: int | None = None, pairs: bool = True,
                                 expansion: str = "auto", out: dict | None = None) -> dict:
        """:meth:`trace_beam_pruned` with STATIC output shapes and no host synchronisation
        (``drt_trace_paths_beam_async``): the form a ``jax.ffi`` handler or a HIP graph needs (reference boundary:
        ``wp.jax_callable(func, output_dims=...)``, _mesh.py:266-276).  Returns device tensors ``keys [max_paths]``,
        ``vertices [max_paths, order+2, 3]``, ``objects [max_paths, order+2]`` -- valid paths first, in
        ``masked_vertices`` order, then padding (key -1) -- and ``counts [4]`` (``[1]`` valid paths, ``[2]`` status word:
        non-zero = a capacity overflowed, re-run larger or use :meth:`trace_beam_pruned`).  Pass the returned dict
        back as ``out`` to reuse every buffer (capture + replay).  The mesh's primitive clusters (and its LBVH with
        ``accel="bvh"``) are built here, outside any capture, on first use.  ``expansion="fused"``: the last expansion as one
        kernel (``DRT_BEAM_EXPAND_FUSED``: cross-check and A/B of the two-kernel default; same rows)."""
        if not 0 <= order <= 3:
            raise ValueError("beam pruning covers orders 0..3")
        if expansion not in ("auto", "clustered", "fused"):
            raise ValueError("expansion must be 'auto', 'clustered' or 'fused'")
        beam = _lib.BeamParams()
        beam.kappa = float(kappa)
        beam.flags = (0 if pairs else _lib.DRT_BEAM_NO_PAIRS) | (_lib.DRT_BEAM_EXPAND_FUSED if expansion == "fused" else 0)
        beam.max_entries, beam.max_records = int(max_entries or 0), int(max_records or 0)
        beam.max_rows, beam.max_survivors = int(max_rows or 0), int(max_survivors or 0)
        tx = scene.transmitters.reshape(-1, 3).contiguous()
        rx = scene.receivers.reshape(-1, 3).contiguous()
        mesh = scene.mesh
        dev = tx.device
        h = mesh.handle().h
        params = _params(self.epsilon, self.hit_tol, self.min_len, self.accel)
        if out is None:
            lib = _lib.load()
            if mesh.num_primitives:
                # (the handle keeps both kinds of clusters once built, each at a fixed address: a captured graph stays valid)
                _lib.call("drt_mesh_build_beam_clusters_ex", h, 1 if pairs else 0, stream())
                if self.accel == "bvh":
                    _lib.call("drt_mesh_build_bvh", h, stream())
            nbytes = lib.drt_trace_beam_workspace_size(tx.shape[0], rx.shape[0], mesh.num_primitives, order, C.byref(beam),
                                                       int(max_paths))
            out = {"keys": torch.empty(max_paths, dtype=torch.int64, device=dev),
                   "vertices": torch.empty((max_paths, order + 2, 3), dtype=torch.float32, device=dev),
                   "objects": torch.empty((max_paths, order + 2), dtype=torch.int32, device=dev),
                   "counts": torch.zeros(4, dtype=torch.int64, device=dev),
                   "workspace": torch.empty(nbytes, dtype=torch.uint8, device=dev)}
        else:  # reused buffers: the C side writes max_paths rows into whatever it is given
            need = _lib.load().drt_trace_beam_workspace_size(tx.shape[0], rx.shape[0], mesh.num_primitives, order,
                                                             C.byref(beam), int(max_paths))
            want = {"keys": ((max_paths,), torch.int64), "vertices": ((max_paths, order + 2, 3), torch.float32),
                    "objects": ((max_paths, order + 2), torch.int32), "counts": ((4,), torch.int64)}
            for name, (shape, dtype) in want.items():
                t = out[name]
                if tuple(t.shape) != shape or t.dtype != dtype or t.device != dev or not t.is_contiguous():
                    raise ValueError(f"out[{name!r}] must be a contiguous {dtype} tensor of shape {shape} on {dev}, "
                                     f"got {tuple(t.shape)} {t.dtype} {t.device}")
            if out["workspace"].numel() < need or out["workspace"].device != dev:
                raise ValueError(f"out['workspace'] holds {out['workspace'].numel()} bytes, this call needs {need}")
        ws = out["workspace"]
        _lib.call("drt_trace_paths_beam_async", h, C.byref(params), C.byref(beam), ptr(tx), tx.shape[0], ptr(rx),
                  rx.shape[0], order, int(max_paths), ptr(out["keys"]), ptr(out["vertices"]), ptr(out["objects"]),
                  ptr(out["counts"]), ptr(ws), ws.numel(), stream())
        return out

    def _beam_workspace(self, nbytes: int, dev) -> torch.Tensor:
        """One workspace per tracer, grown on demand and reused from call to call (the default list capacities
        add up to a few GiB: allocating them per call would dominate small scenes)."""
        ws = getattr(self, "_beam_ws", None)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            self._beam_ws = ws = None  # release before growing
            self._beam_ws = ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    def _trace_compact(self, scene, desc, max_survivors, max_paths) -> TracedPaths:
        tx = scene.transmitters.reshape(-1, 3).contiguous()
        rx = scene.receivers.reshape(-1, 3).contiguous()
        params = _params(self.epsilon, self.hit_tol, self.min_len, self.accel,
                         skip_occlusion=bool(getattr(self, "_skip_occlusion", False)),
                         deterministic_grad=self.deterministic_grad)
        st = None
        if self.collect_stats:
            st = _lib.TraceStats()
            params.stats = C.pointer(st)
        verts, objs, keys = _TraceCompactFn.apply(
            tx, rx, scene.mesh.vertices, scene.mesh, desc, params, max_survivors, max_paths,
        )
        if st is not None:
            self.last_stats = {f: getattr(st, f) for f, _ in _lib.TraceStats._fields_ if f != "reserved"}
        n, order = objs.shape[0], desc["order"]
        return TracedPaths(
            verts, objs, torch.ones(n, dtype=torch.bool, device=objs.device),
            torch.zeros((n, order), dtype=torch.int32, device=objs.device),
            self.confidence_threshold, keys,
        )


@dataclass
class HybridPathTracer(ExhaustivePathTracer):
    """Visibility-pruned exhaustive tracer (reference _solvers.py:960-1176): ray launching estimates
    which primitives are visible from the transmitters / receivers, the first (last) interaction of
    a candidate is restricted to them, then candidates are traced exhaustively.

    Visibility is merged over all transmitters (receivers), as in the reference (:969-973)."""

    num_rays: int = int(1e6)
    sample_triangles: bool = False
    """Extension (needs ``accel="bvh"``): complement the lattice visibility estimate with interior sample points
    of every face (``Mesh.triangles_visible_from_vertex(sample_triangles=True)``)."""
    pairs_strategy: str = "auto"
    """``"auto"``, ``"ragged"`` (one lane per (pair, candidate) row), ``"prefix"`` (order >= 3: one lane per
    first ``order - 1`` interactions, inner loops over receivers and last interactions) or ``"loop"`` (one
    product-space launch per pair).  ``"auto"`` takes the prefix kernel at order >= 3 above a mean of 2e7 rows per pair
    (decided inside ``drt_trace_paths_hybrid_pairs``, csrc/hybrid.hip, where the row count is known)."""

    def _graph(self, scene):
        mesh = scene.mesh
        tx = scene.transmitters.reshape(-1, 3)
        rx = scene.receivers.reshape(-1, 3)
        vis_tx = mesh.triangles_visible_from_vertex(tx, num_rays=self.num_rays, accel=self.accel, sample_triangles=self.sample_triangles).any(dim=0)
        vis_rx = mesh.triangles_visible_from_vertex(rx, num_rays=self.num_rays, accel=self.accel, sample_triangles=self.sample_triangles).any(dim=0)
        if mesh.assume_quads:  # _solvers.py:1024-1031
            vis_tx = vis_tx.reshape(-1, 2).any(dim=-1)
            vis_rx = vis_rx.reshape(-1, 2).any(dim=-1)
        graph = DiGraph.from_complete_graph(CompleteGraph(mesh.num_primitives))
        from_, to = graph.insert_from_and_to_nodes(
            from_adjacency=vis_tx.cpu().numpy(), to_adjacency=vis_rx.cpu().numpy()
        )
        if mesh.mask is not None:  # _solvers.py:1038-1042
            mask = mesh.mask
            if mesh.assume_quads:
                mask = mask[0::2] & mask[1::2]
            graph.filter_by_mask(mask.cpu().numpy(), fast_mode=True)
        return graph, from_, to

    def _fast_candidates(self, scene, order):
        """Orders 1 and 2 of the pruned DiGraph (_solvers.py:1013-1056) without building it on the host: a path
        from -> n_1 -> .. -> n_k -> to exists iff n_1 is visible from a transmitter, n_k from a receiver, every n_i is
        active and consecutive nodes differ (a complete graph has no self loops) -- enumerated in the DiGraph's
        lexicographic order (graph.rs:400-470) with two torch ops on the device.  Higher orders: the host iterator."""
        if order not in (1, 2):
            return None
        first, last, _middle, both = self._visible_sets(scene)
        scale = 2 if scene.mesh.assume_quads else 1
        if order == 1:
            return (torch.nonzero(both).reshape(-1).to(torch.int32) * scale).reshape(-1, 1).contiguous()
        if first.shape[0] * last.shape[0] >= 2 ** 27:
            return None
        rows = torch.cartesian_prod(first, last).reshape(-1, 2)  # (first-major: lexicographic)
        return (rows[rows[:, 0] != rows[:, 1]] * scale).contiguous()

    def _visible_sets(self, scene):
        """Primitive index lists (device int32, ascending): first interactions (visible from a
        transmitter), last interactions (visible from a receiver), and the middle set (all active
        primitives, ``None`` = every primitive) -- the node sets of the reference's pruned DiGraph
        (_solvers.py:1013-1042)."""
        mesh = scene.mesh
        tx = scene.transmitters.reshape(-1, 3)
        rx = scene.receivers.reshape(-1, 3)
        # num_path_candidates() followed by trace_rank_range() (e.g. trace_rank_range_sharded) would otherwise
        # launch the visibility rays twice: memoise the last result on everything it depends on
        # (end points by VALUE: a fresh tensor of an optimisation step may reuse the previous one's memory)
        def key():
            # the handle object itself (kept alive by the cache entry): a re-snapshot makes a new one
            return (mesh.handle(), self.num_rays, self.accel, self.sample_triangles,
                    tx.detach().cpu().numpy().tobytes(), rx.detach().cpu().numpy().tobytes())

        cached = getattr(self, "_vis_cache", None)
        if cached is not None and cached[0] == key():
            return cached[1]
        out = self._visible_sets_uncached(mesh, tx, rx)
        self._vis_cache = (key(), out)  # after the call: the first query may move the mesh to the device
        return out

    def _visible_sets_uncached(self, mesh, tx, rx):
        vis_tx = mesh.triangles_visible_from_vertex(tx, num_rays=self.num_rays, accel=self.accel, sample_triangles=self.sample_triangles).any(dim=0)
        vis_rx = mesh.triangles_visible_from_vertex(rx, num_rays=self.num_rays, accel=self.accel, sample_triangles=self.sample_triangles).any(dim=0)
        if mesh.assume_quads:
            vis_tx = vis_tx.reshape(-1, 2).any(dim=-1)
            vis_rx = vis_rx.reshape(-1, 2).any(dim=-1)
        middle = None
        if mesh.mask is not None:
            active = mesh.mask
            if mesh.assume_quads:
                active = active[0::2] & active[1::2]
            vis_tx, vis_rx = vis_tx & active, vis_rx & active
            middle = torch.nonzero(active).reshape(-1).to(torch.int32).contiguous()

        def ids(b):
            return torch.nonzero(b).reshape(-1).to(torch.int32).contiguous()

        return ids(vis_tx), ids(vis_rx), middle, (vis_tx & vis_rx)

    def estimate_visibility(self, scene) -> tuple[torch.Tensor, torch.Tensor]:
        """``(bool[num_tx, T], bool[num_rx, T])``: triangles seen from every transmitter / receiver
        (``num_rays`` lattice rays each, first hit on the LBVH).  Reusable across :meth:`trace_pairs` calls
        while the end points move little (e.g. the steps of a gradient descent on TX positions)."""
        mesh = scene.mesh
        tx = scene.transmitters.reshape(-1, 3)
        rx = scene.receivers.reshape(-1, 3)
        return (mesh.triangles_visible_from_vertex(tx, num_rays=self.num_rays, accel=self.accel, sample_triangles=self.sample_triangles),
                mesh.triangles_visible_from_vertex(rx, num_rays=self.num_rays, accel=self.accel, sample_triangles=self.sample_triangles))

    def trace_pairs(self, scene, order: int, *, visibility: tuple[torch.Tensor, torch.Tensor] | None = None,
                    max_survivors: int = 1 << 20, max_paths: int = 1 << 14) -> TracedPaths:
        """MI355X extension: visibility pruning PER (transmitter, receiver) pair instead of merged over
        all of them (the reference merges, _solvers.py:969-973, which prunes little once there are many
        end points): pair (i, j) traces ``F_i x N^(order-2) x L_j`` -- first interaction visible from
        transmitter i, last one from receiver j -- all pairs in ONE launch over the concatenated (ragged) spaces.  Returns the valid
        paths of all pairs (pair-major, lexicographic inside a pair = ``masked_vertices`` order of the
        exhaustive tracer), differentiable like any compact trace.  Finds a subset of the exhaustive
        tracer's valid paths that is complete up to the sampling of the visibility estimate;
        ``visibility`` takes a cached :meth:`estimate_visibility` result.  ``keys`` are packed like those of
        :meth:`trace_beam_pruned`: ``(tx*num_rx + rx) * n**order + sum_j m_j * n**(order-1-j)``."""
        if order < 2:
            return self.trace_rank_range(scene, order, max_survivors=max_survivors, max_paths=max_paths)
        mesh = scene.mesh
        tx = scene.transmitters.reshape(-1, 3).contiguous()
        rx = scene.receivers.reshape(-1, 3).contiguous()
        vis_tx, vis_rx = self.estimate_visibility(scene) if visibility is None else visibility
        # a cached estimate must be THIS scene's: the native call indexes vis[v * T + f] unconditionally
        for name, vis, pts in (("transmitters", vis_tx, tx), ("receivers", vis_rx, rx)):
            want = (pts.shape[0], mesh.num_triangles)
            if not isinstance(vis, torch.Tensor) or tuple(vis.shape) != want or vis.device != tx.device \
                    or vis.dtype not in (torch.bool, torch.uint8):
                raise ValueError(f"visibility of the {name} must be a bool / uint8 tensor of shape {want} on {tx.device} "
                                 f"(estimate_visibility of this scene), got "
                                 f"{tuple(vis.shape) if isinstance(vis, torch.Tensor) else type(vis).__name__}")
        strategy = self.pairs_strategy
        if strategy == "loop":  # one product-space launch per pair (a debugging mapping; torch glue)
            vt, vr = vis_tx, vis_rx
            if mesh.assume_quads:
                vt = vt.reshape(vt.shape[0], -1, 2).any(dim=-1)
                vr = vr.reshape(vr.shape[0], -1, 2).any(dim=-1)
            middle = None
            if mesh.mask is not None:
                active = mesh.mask
                if mesh.assume_quads:
                    active = active[0::2] & active[1::2]
                vt, vr = vt & active, vr & active
                middle = torch.nonzero(active).reshape(-1).to(torch.int32).contiguous()
            n = mesh.num_primitives if middle is None else int(middle.shape[0])
            return self._trace_pairs_loop(scene, order, vt, vr, middle, n, max_survivors, max_paths)
        if strategy not in ("auto", "ragged", "prefix"):
            raise ValueError(f"unknown pairs_strategy {self.pairs_strategy!r}")
        # everything else -- primitive-level visibility, CSR sets, pair offsets, the ragged trace -- is ONE native call
        # (drt_trace_paths_hybrid_pairs, csrc/hybrid.hip); this method only marshals
        flags = {"auto": 0, "ragged": _lib.DRT_HYBRID_RAGGED, "prefix": _lib.DRT_HYBRID_PREFIX}[strategy]
        if mesh.num_primitives and tx.shape[0] * rx.shape[0] * mesh.num_primitives ** order >= 2 ** 62:
            raise OverflowError("tx * rx * primitives**order does not fit a 62-bit key")
        info: dict = {}
        params = _params(self.epsilon, self.hit_tol, self.min_len, self.accel, deterministic_grad=self.deterministic_grad)
        verts, objs, keys = _TraceHybridPairsFn.apply(
            tx, rx, mesh.vertices, mesh, order, params, vis_tx.to(torch.uint8).contiguous(), vis_rx.to(torch.uint8).contiguous(),
            flags, int(max_survivors), int(max_paths), info)
        self.last_num_evaluated = info.get("evaluated", 0)
        nv = objs.shape[0]
        return TracedPaths(verts, objs, torch.ones(nv, dtype=torch.bool, device=objs.device),
                           torch.zeros((nv, order), dtype=torch.int32, device=objs.device), self.confidence_threshold, keys)

    def _trace_pairs_loop(self, scene, order, vis_tx, vis_rx, middle, n, max_survivors, max_paths) -> TracedPaths:
        """One GPU-unranked product-space launch per (transmitter, receiver) pair."""
        tx = scene.transmitters.reshape(-1, 3)
        rx = scene.receivers.reshape(-1, 3)
        mesh = scene.mesh
        firsts = [torch.nonzero(v).reshape(-1).to(torch.int32).contiguous() for v in vis_tx]
        lasts = [torch.nonzero(v).reshape(-1).to(torch.int32).contiguous() for v in vis_rx]
        from ._scene import Scene

        verts, objs, evaluated = [], [], 0
        for i, f in enumerate(firsts):
            for j, l in enumerate(lasts):
                total = int(f.shape[0]) * n ** (order - 2) * int(l.shape[0])
                evaluated += total
                if total == 0:
                    continue
                sub = Scene(tx[i:i + 1], rx[j:j + 1], mesh)
                desc = {"table": None, "order": order, "rank_lo": 0, "count": total, "num_nodes": n,
                        "node_map": middle, "first_map": f, "last_map": l}
                p = self._trace_compact(sub, desc, max_survivors, max_paths)
                if p.objects.shape[0]:
                    o = p.objects.clone()
                    o[:, 0], o[:, -1] = i, j
                    verts.append(p.vertices)
                    objs.append(o)
        self.last_num_evaluated = evaluated
        dev = tx.device
        if not verts:
            return TracedPaths(torch.zeros((0, order + 2, 3), device=dev), torch.zeros((0, order + 2), dtype=torch.int32, device=dev),
                               torch.zeros(0, dtype=torch.bool, device=dev), torch.zeros((0, order), dtype=torch.int32, device=dev),
                               self.confidence_threshold)
        v, o = torch.cat(verts), torch.cat(objs)
        return TracedPaths(v, o, torch.ones(o.shape[0], dtype=torch.bool, device=dev),
                           torch.zeros((o.shape[0], order), dtype=torch.int32, device=dev), self.confidence_threshold)


    def num_path_candidates(self, scene, order: int) -> int:
        """Size of the pruned rank space of :meth:`trace_rank_range` (for order >= 2 it still counts the
        tuples with two equal neighbours, which are skipped while tracing)."""
        first, last, middle, both = self._visible_sets(scene)
        if order == 0:
            return 1
        if order == 1:
            return int(both.sum())
        n = scene.mesh.num_primitives if middle is None else int(middle.shape[0])
        return int(first.shape[0]) * n ** (order - 2) * int(last.shape[0])

    def trace_rank_range(self, scene, order: int, rank_lo: int = 0, rank_hi: int | None = None, *,
                         max_survivors: int = 1 << 20, max_paths: int = 1 << 16, literal: bool | None = None) -> TracedPaths:
        """GPU-resident counterpart of the pruned DiGraph enumeration (_solvers.py:996-1056): ranks
        address ``F x N^(order-2) x L`` (first interaction visible from a transmitter, last one from a
        receiver, inactive primitives removed) in lexicographic order -- the DFS order of the
        reference -- and are unranked inside the trace kernels; no candidate table, no host DFS.
        Valid paths come back in ``masked_vertices`` order, identical to tracing the host-generated
        table (``tests/test_trace_gpu.py``)."""
        if self.smoothing_factor is not None:
            raise NotImplementedError("the smoothed mode is dense by nature: use trace_path_candidates")
        first, last, middle, both = self._visible_sets(scene)
        if order == 0:
            return ExhaustivePathTracer.trace_rank_range(self, scene, 0, rank_lo, rank_hi,
                                                         max_survivors=max_survivors, max_paths=max_paths)
        if order == 1:  # the single interaction must be visible from both ends
            node_map = torch.nonzero(both).reshape(-1).to(torch.int32).contiguous()
            total = int(node_map.shape[0])
            hi = total if rank_hi is None else min(int(rank_hi), total)
            lo = min(int(rank_lo), hi)
            desc = {"table": None, "order": 1, "rank_lo": lo, "count": hi - lo, "num_nodes": max(total, 1),
                    "node_map": node_map}
            return self._trace_compact(scene, desc, max_survivors, max_paths)
        n = scene.mesh.num_primitives if middle is None else int(middle.shape[0])
        total = int(first.shape[0]) * n ** (order - 2) * int(last.shape[0])
        hi = total if rank_hi is None else min(int(rank_hi), total)
        lo = min(int(rank_lo), hi)
        desc = {"table": None, "order": order, "rank_lo": lo, "count": hi - lo, "num_nodes": n, "node_map": middle,
                "first_map": first, "last_map": last}
        return self._trace_compact(scene, desc, max_survivors, max_paths)


class _LaunchPathsFn(torch.autograd.Function):
    """Bounce points of the fused SBR kernel, differentiable in (ray origins, ray directions, mesh
    vertices) through ``drt_launch_paths_vjp`` (reference: launch_paths is plain differentiable JAX
    around ``Mesh.first_triangle_hit_by_ray``, _solvers.py:385-444)."""

    @staticmethod
    def forward(ctx, ro, rd, mesh_vertices, mesh, rx, order, eps, max_dist):
        ntx, R = ro.shape[0], ro.shape[1]
        nrx, dev = rx.shape[0], ro.device
        tris = torch.empty((ntx, R, order), dtype=torch.int32, device=dev)
        verts = torch.empty((ntx, R, order, 3), dtype=torch.float32, device=dev)
        masks = torch.zeros((ntx, nrx, R, order + 1), dtype=torch.uint8, device=dev)
        _lib.call("drt_launch_paths", mesh.handle().h, ptr(ro), ptr(rd), ntx, R, ptr(rx), nrx,
                  order, eps, 512, max_dist, ptr(tris), ptr(verts), ptr(masks), stream())
        ctx.mesh, ctx.cfg = mesh, (order, eps)
        ctx.save_for_backward(ro, rd, tris)
        ctx.mark_non_differentiable(tris, masks)
        return verts, tris, masks

    @staticmethod
    def backward(ctx, gv, _gt, _gm):
        ro, rd, tris = ctx.saved_tensors
        order, eps = ctx.cfg
        mesh = ctx.mesh
        need = ctx.needs_input_grad
        go = torch.zeros_like(ro) if need[0] else None
        gd = torch.zeros_like(rd) if need[1] else None
        gmv = torch.zeros_like(mesh.vertices) if need[2] else None
        if ro.numel():
            _lib.call("drt_launch_paths_vjp", mesh.handle().h, ptr(ro), ptr(rd), ro.shape[0], ro.shape[1], order,
                      eps, ptr(tris), ptr(gv.contiguous()), ptr(go), ptr(gd), ptr(gmv), stream())
        return go, gd, gmv, None, None, None, None, None


class AbstractPathLauncher:
    """Ray-launching solver interface of the reference (_solvers.py:250-491): subclasses provide
    ``launch_rays``; ``launch_paths`` runs first-hit / ``filter_rays`` / ``bounce_rays`` for
    ``order + 1`` bounces -- here one fused HIP kernel (csrc/launch.hip) over the mesh LBVH."""

    max_dist: float = 1e-3
    epsilon: float | None = None

    def launch_rays(self, scene):
        raise NotImplementedError

    def launch_paths(self, scene, order: int) -> LaunchedPaths:
        tx = scene.transmitters.reshape(-1, 3).contiguous()
        rx = scene.receivers.reshape(-1, 3).contiguous()
        ro, rd = self.launch_rays(scene)
        ro, rd = ro.contiguous(), rd.contiguous()
        ntx, nrx, R = tx.shape[0], rx.shape[0], ro.shape[1]
        dev = tx.device
        eps = 10.0 * F32_EPS if self.epsilon is None else float(self.epsilon)
        verts, tris, masks = _LaunchPathsFn.apply(ro, rd, scene.mesh.vertices, scene.mesh, rx.detach(), order, eps,
                                                  float(self.max_dist))
        # reference layout [num_tx, num_rx, num_rays, ...] (_solvers.py:446-490): broadcast views
        inner = verts[:, None].expand(ntx, nrx, R, order, 3)
        vertices = torch.cat((tx[:, None, None, None, :].expand(ntx, nrx, R, 1, 3), inner,
                              rx[None, :, None, None, :].expand(ntx, nrx, R, 1, 3)), dim=-2)
        ar_tx = torch.arange(ntx, dtype=torch.int32, device=dev)[:, None, None, None].expand(ntx, nrx, R, 1)
        ar_rx = torch.arange(nrx, dtype=torch.int32, device=dev)[None, :, None, None].expand(ntx, nrx, R, 1)
        objects = torch.cat((ar_tx, tris[:, None].expand(ntx, nrx, R, order), ar_rx), dim=-1)
        it = torch.zeros((ntx, nrx, R, order), dtype=torch.int32, device=dev)
        return LaunchedPaths(vertices, objects, masks.bool(), it)


@dataclass
class SBRPathLauncher(AbstractPathLauncher):
    """Shooting-and-bouncing rays (reference _solvers.py:1179-1226), same fields and defaults."""

    num_rays: int = int(1e6)
    epsilon: float | None = None
    hit_tol: float | None = None
    max_dist: float = 1e-3

    def launch_rays(self, scene):
        """A Fibonacci lattice inside the frustum that contains the mesh and the receivers, per
        transmitter (_solvers.py:1202-1226)."""
        tx = scene.transmitters.reshape(-1, 3).contiguous()
        rx = scene.receivers.reshape(-1, 3)
        world = torch.cat((scene.mesh.triangle_vertices.detach().reshape(-1, 3), rx)).contiguous()
        ntx = tx.shape[0]
        fr = torch.empty((ntx, 2, 3), dtype=torch.float32, device=tx.device)
        _lib.call("drt_viewing_frustum_points", ptr(tx), ntx, ptr(world), world.shape[0], ptr(fr), stream())
        dirs = torch.empty((ntx, self.num_rays, 3), dtype=torch.float32, device=tx.device)
        for i in range(ntx):
            _lib.call("drt_fibonacci_lattice", self.num_rays, ptr(fr[i]), ptr(dirs[i]), stream())
        return tx[:, None, :].expand(ntx, self.num_rays, 3).contiguous(), dirs
