"""Multi-GPU execution of the hot path: one process per GPU (``torch.distributed``; backend
``nccl`` = RCCL over xGMI on MI355X, ``gloo`` in the CPU tests).

The path shards naturally: the lexicographic candidate-rank space
``[0, C)`` (or the ray axis of a stand-alone ray query) is cut into one contiguous block per rank,
the mesh (a few MB) is replicated, and NO collective runs during compute.  The epilogue is

* ``gather_paths``      -- one ``all_gather`` of per-rank counts, then one of padded records, then a
                           sort of the (few) valid paths by their global flat key: the result is the
                           single-GPU order of ``TracedPaths.masked_vertices`` on every rank;
* ``allreduce_grads``   -- one SUM all-reduce of the [N_tx,3] (+[N_rx,3], +[N_v,3]) gradients;
* ``trace_beam_pruned_sharded`` -- the conservatively pruned full-coverage search, split by
                           (transmitter, first mirror) prefix; same epilogue;
* ``reduce_first_hit``  -- triangle-block sharding (BASELINE configs[4]): every rank holds a block
                           of triangles and produces packed ``(t, tie)`` 64-bit keys for the same
                           rays; a MIN all-reduce picks the global first hit with the reference's
                           tile tie-break, independent of the partition.
"""

from __future__ import annotations

import torch
import torch.distributed as dist

__all__ = [
    "allreduce_grads",
    "reduce_any_hit",
    "trace_rank_range_triangle_sharded",
    "first_triangle_hit_by_ray_sharded",
    "gather_paths",
    "globalize_keys",
    "reduce_first_hit",
    "shard_interval",
    "trace_rank_range_sharded",
    "trace_beam_pruned_sharded",
    "NativeComm",
    "native_comm",
]


def shard_interval(total: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of ``range(total)`` owned by ``rank`` (sizes differ by <= 1)."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad world_size / rank")
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class NativeComm:
    """``drt_comm_t``: the path's collectives straight on RCCL (csrc/comm.hip), no torch.distributed in the data
    path.  ``unique_id`` (128 bytes from :meth:`unique_id` on rank 0) reaches the other ranks out of band."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        import ctypes as C

        from . import _lib

        self._lib, self.h = _lib, C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.call("drt_comm_init", buf, int(rank), int(world), C.byref(self.h))
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib

        buf = (C.c_uint8 * 128)()
        _lib.call("drt_comm_unique_id", buf)
        return bytes(buf)

    def close(self) -> None:
        if getattr(self, "h", None):
            self._lib.call("drt_comm_destroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _call(self, name: str, t: torch.Tensor) -> torch.Tensor:
        from ._tensors import ptr, stream

        assert t.is_contiguous()
        self._lib.call(name, self.h, ptr(t), t.numel(), stream())
        return t

    def allreduce_min_u64(self, keys_i64: torch.Tensor) -> torch.Tensor:
        """In place; the tensor holds UNSIGNED 64-bit keys in an int64 container."""
        return self._call("drt_allreduce_min_u64", keys_i64)

    def allreduce_max_u8(self, flags: torch.Tensor) -> torch.Tensor:
        return self._call("drt_allreduce_max_u8", flags)

    def allreduce_sum_f32(self, x: torch.Tensor) -> torch.Tensor:
        return self._call("drt_allreduce_sum_f32", x)

    def allgather_bytes(self, send: torch.Tensor) -> torch.Tensor:
        from ._tensors import ptr, stream

        send = send.contiguous()
        nbytes = send.numel() * send.element_size()
        recv = torch.empty((self.world, *send.shape), dtype=send.dtype, device=send.device)
        self._lib.call("drt_allgather_bytes", self.h, ptr(send), ptr(recv), nbytes, stream())
        return recv


_NATIVE: dict = {}  # process group OBJECT -> NativeComm (the key keeps the group alive: its identity cannot be reused)


def native_comm(group=None) -> NativeComm | None:
    """The native communicator of ``group`` when ``DRT_COMM=rccl`` (created on first use: rank 0 draws the id,
    torch.distributed's broadcast carries its 128 bytes -- rendezvous only, never the data path)."""
    import os

    if os.environ.get("DRT_COMM", "").lower() != "rccl":
        return None
    world, rank = _world(group)
    if world == 1:
        return None
    key = group if group is not None else dist.group.WORLD
    nc = _NATIVE.get(key)
    if nc is not None and (nc.world != world or nc.rank != rank or nc.h is None):
        nc.close()  # a re-initialised default group: never hand out a communicator of another world
        nc = None
    if nc is None:
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt = torch.tensor(list(NativeComm.unique_id()), dtype=torch.uint8, device=dev)
        dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        nc = _NATIVE[key] = NativeComm(bytes(idt.cpu().tolist()), rank, world)
    return nc


def _world(group) -> tuple[int, int]:
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _allgather_rows(t: torch.Tensor, group) -> torch.Tensor:
    """``[world, *t.shape]``: ONE collective -- ``drt_allgather_bytes`` under ``DRT_COMM=rccl`` on GPU tensors, else
    ``torch.distributed`` (into one tensor where the backend can, a list otherwise)."""
    world, _ = _world(group)
    t = t.contiguous()
    nc = native_comm(group)
    if nc is not None and t.is_cuda:
        return nc.allgather_bytes(t)
    out = torch.empty((world, *t.shape), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(out, t, group=group)
    except (RuntimeError, NotImplementedError):  # a backend without the flat form
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t, group=group)
        out = torch.stack(bufs)
    return out


def gather_paths(keys: torch.Tensor, vertices: torch.Tensor, objects: torch.Tensor,
                 key_offset: int = 0, group=None):
    """All-gather the valid paths of every rank; returns ``(keys, vertices, objects)`` identical on
    all ranks, sorted by global flat key ``(tx*num_rx + rx) * C + candidate_rank`` (the order of
    ``masked_vertices``).  ``keys`` must already be global (see ``globalize_keys``); ``key_offset``
    is a convenience for the single-pair case.

    TWO collectives: the per-rank counts (8 bytes each; their maximum sizes the record buffer -- the one host
    read of the epilogue), then every rank's paths as ONE block of packed records ``key | vertices | objects``
    (8 + 16 (order + 2) bytes per path) padded to that maximum."""
    world, _ = _world(group)
    keys = keys + key_offset
    if world == 1:
        return keys, vertices, objects
    dev = keys.device
    n = keys.shape[0]
    counts = _allgather_rows(torch.tensor([n], dtype=torch.int64, device=dev), group).reshape(-1)
    counts_h = counts.tolist()
    cap = max(max(counts_h), 1)
    k2 = vertices.shape[1]
    rec = 8 + 12 * k2 + 4 * k2
    block = torch.zeros((cap, rec), dtype=torch.uint8, device=dev)
    if n:
        block[:n, :8] = keys.contiguous().view(torch.uint8).reshape(n, 8)
        block[:n, 8:8 + 12 * k2] = vertices.contiguous().view(torch.uint8).reshape(n, 12 * k2)
        block[:n, 8 + 12 * k2:] = objects.contiguous().view(torch.uint8).reshape(n, 4 * k2)
    allb = _allgather_rows(block, group)  # [world, cap, rec]
    valid = (torch.arange(cap, device=dev)[None, :] < counts[:, None]).reshape(-1)
    rows = allb.reshape(world * cap, rec)[valid].contiguous()
    m = rows.shape[0]
    out_k = rows[:, :8].contiguous().view(torch.int64).reshape(m)
    out_v = rows[:, 8:8 + 12 * k2].contiguous().view(torch.float32).reshape(m, k2, 3)
    out_o = rows[:, 8 + 12 * k2:].contiguous().view(torch.int32).reshape(m, k2)
    perm = torch.argsort(out_k, stable=True)
    return out_k[perm], out_v[perm], out_o[perm]


def globalize_keys(local_keys: torch.Tensor, local_count: int, rank_lo: int, total: int) -> torch.Tensor:
    """Keys of a rank-window trace ``pair * local_count + (rank - rank_lo)`` -> global flat keys
    ``pair * total + rank``."""
    pair = torch.div(local_keys, local_count, rounding_mode="floor")
    return pair * total + rank_lo + (local_keys - pair * local_count)


def allreduce_grads(*grads: torch.Tensor, group=None) -> None:
    """In-place SUM all-reduce of gradient tensors (one flat bucket, one collective)."""
    world, _ = _world(group)
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    nc = native_comm(group)
    if nc is not None and flat.dtype == torch.float32 and flat.is_cuda:
        nc.allreduce_sum_f32(flat)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].reshape(g.shape))
        off += g.numel()


def reduce_first_hit(packed_keys: torch.Tensor, group=None) -> torch.Tensor:
    """MIN all-reduce of packed first-hit keys ``(ordered(t) << 32) | tie`` held as int64 with the
    sign bit flipped (so that signed MIN == unsigned MIN; RCCL has no uint64 MIN on every build)."""
    world, _ = _world(group)
    if world > 1:
        nc = native_comm(group)
        if nc is not None and packed_keys.is_cuda:
            # the native path reduces UNSIGNED keys: undo / redo the sign-bit flip of the torch path around it
            flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=packed_keys.device)
            packed_keys.bitwise_xor_(flip)
            nc.allreduce_min_u64(packed_keys)
            packed_keys.bitwise_xor_(flip)
        else:
            dist.all_reduce(packed_keys, op=dist.ReduceOp.MIN, group=group)
    return packed_keys


def reduce_any_hit(blocked: torch.Tensor, group=None) -> torch.Tensor:
    """MAX all-reduce of per-path ``blocked`` flags (uint8): a path is occluded if ANY rank's triangle block
    occludes one of its segments."""
    world, _ = _world(group)
    if world > 1:
        nc = native_comm(group)
        if nc is not None and blocked.is_cuda and blocked.dtype == torch.uint8:
            nc.allreduce_max_u8(blocked)
        else:
            dist.all_reduce(blocked, op=dist.ReduceOp.MAX, group=group)
    return blocked


def trace_rank_range_triangle_sharded(tracer, scene, order: int, rank_lo: int = 0, rank_hi: int | None = None, *,
                                      tri_lo: int | None = None, tri_hi: int | None = None, group=None, **kwargs):
    """Triangle-block sharding of the tracer's OCCLUSION stage (BASELINE configs[4], scenes whose triangle data
    should not be tested whole on one GPU): every rank runs the geometric stage on the same candidates (the
    mirror planes of a candidate can be any triangle, so vertices / normals stay replicated -- 48 B per
    triangle), tests the k+1 segments of each survivor against ITS block ``[tri_lo, tri_hi)`` only
    (default: ``shard_interval(T, world, rank)``) and ONE MAX all-reduce of a byte per survivor decides.
    Returns the valid paths as a ``TracedPaths`` identical to the unsharded ``trace_rank_range`` (the
    occlusion predicate is an OR over triangles, so any partition gives the same mask)."""
    from .geometry._paths import TracedPaths
    from .geometry._utils import ray_intersect_any_triangle

    world, rank = _world(group)
    mesh = scene.mesh
    T = mesh.num_triangles
    if tri_lo is None or tri_hi is None:
        tri_lo, tri_hi = shard_interval(T, world, rank)
    tracer._skip_occlusion = True  # geometric survivors only (DRT_TRACE_SKIP_OCCLUSION)
    try:
        p = tracer.trace_rank_range(scene, order, rank_lo, rank_hi, literal=True, **kwargs)  # (the filter stage itself, not the pruned search)
    finally:
        tracer._skip_occlusion = False
    S = p.objects.shape[0]
    blocked = torch.zeros(S, dtype=torch.uint8, device=p.objects.device)
    if S and tri_hi > tri_lo:
        v = p.vertices.detach()
        o = v[:, :-1, :].reshape(-1, 3)
        d = (v[:, 1:, :] - v[:, :-1, :]).reshape(-1, 3)
        tvb = mesh.triangle_vertices.detach()[tri_lo:tri_hi].contiguous()
        act = None if mesh.mask is None else mesh.mask[tri_lo:tri_hi].contiguous()
        hit = ray_intersect_any_triangle(o, d, tvb, act, hit_tol=tracer.hit_tol, epsilon=tracer.epsilon)
        blocked = hit.reshape(S, order + 1).any(dim=1).to(torch.uint8)
    reduce_any_hit(blocked, group=group)
    keep = blocked == 0
    n = int(keep.sum().item())
    dev = p.objects.device
    return TracedPaths(p.vertices[keep], p.objects[keep], torch.ones(n, dtype=torch.bool, device=dev),
                       torch.zeros((n, order), dtype=torch.int32, device=dev), tracer.confidence_threshold,
                       p.keys[keep])


def trace_rank_range_sharded(tracer, scene, order: int, rank_lo: int = 0, rank_hi: int | None = None,
                             group=None, gather: bool = True, **kwargs):
    """Candidate-rank sharding of ``ExhaustivePathTracer.trace_rank_range``:
    every rank traces one contiguous block of ``[rank_lo, rank_hi)`` against its replica of the mesh,
    no collective during compute.  Returns ``(keys, vertices, objects)`` with GLOBAL keys
    ``(tx*num_rx + rx) * (rank_hi - rank_lo) + (rank - rank_lo)``; gathered (and sorted = the
    single-GPU ``masked_vertices`` order) on every rank when ``gather``, else this rank's part.
    ``vertices`` of the local part stay attached to autograd; call ``allreduce_grads`` on the
    transmitter / receiver gradients after ``backward``."""
    world, rank = _world(group)
    total = tracer.num_path_candidates(scene, order)
    hi = total if rank_hi is None else min(int(rank_hi), total)
    lo = min(int(rank_lo), hi)
    a, b = shard_interval(hi - lo, world, rank)
    local = tracer.trace_rank_range(scene, order, lo + a, lo + b, literal=True, **kwargs)  # candidate-rank sharding IS the exhaustive path
    keys = globalize_keys(local.keys, b - a, a, hi - lo)
    if not gather or world == 1:
        return keys, local.vertices, local.objects
    return gather_paths(keys, local.vertices.detach(), local.objects, group=group)


def trace_beam_pruned_sharded(tracer, scene, order: int, group=None, gather: bool = True, **kwargs):
    """Prefix sharding of ``ExhaustivePathTracer.trace_beam_pruned`` (full coverage of the candidate space
    with the guarantee of DESIGN.md section 9): rank r expands the (transmitter t, first mirror m) prefixes
    with (t * n + m) % world == r against its replica of the mesh -- no collective during compute; the epilogue is the
    one of the exhaustive sharding (``gather_paths``: keys are already global).  Returns ``(keys, vertices,
    objects)``: gathered and sorted (= the single-GPU result, bit for bit) on every rank when ``gather``,
    else this rank's part with ``vertices`` attached to autograd (call ``allreduce_grads`` after
    ``backward``)."""
    world, rank = _world(group)
    local = tracer.trace_beam_pruned(scene, order, prefix_shard=(rank, world) if world > 1 else None, **kwargs)
    if not gather or world == 1:
        return local.keys, local.vertices, local.objects
    return gather_paths(local.keys, local.vertices.detach(), local.objects, group=group)


def first_triangle_hit_by_ray_sharded(ray_origins, ray_directions, triangle_vertices_block,
                                      index_offset: int, total_triangles: int, active_block=None, *,
                                      epsilon: float | None = None, batch_size: int | None = 512,
                                      group=None):
    """Triangle-block sharding of ``first_triangle_hit_by_ray`` (BASELINE configs[4]): this rank holds
    triangles ``[index_offset, index_offset + T_block)`` of a mesh of ``total_triangles``; all ranks
    pass the same rays.  Local packed keys -> one MIN all-reduce (8 B per ray) -> decode.  The result
    (GLOBAL indices, t) equals the single-GPU operator bit for bit, ties included."""
    from . import _lib
    from ._tensors import F32_EPS, as_f32, as_u8, device, ptr, stream

    dev = device()
    o = as_f32(ray_origins, dev).reshape(-1, 3).contiguous()
    d = as_f32(ray_directions, dev).reshape(-1, 3).contiguous()
    tv = as_f32(triangle_vertices_block, dev).reshape(-1, 3, 3).contiguous()
    act = None if active_block is None else as_u8(active_block, dev).reshape(-1).contiguous()
    R = o.shape[0]
    eps = 10.0 * F32_EPS if epsilon is None else float(epsilon)
    bs = 0 if batch_size is None else int(batch_size)
    keys = torch.empty(R, dtype=torch.int64, device=dev)
    _lib.call("drt_first_hit_keys", ptr(o), ptr(d), R, ptr(tv), tv.shape[0], int(index_offset),
              int(total_triangles), ptr(act), eps, bs, ptr(keys), 1, stream())
    # unsigned MIN as a signed MIN: flip the sign bit, reduce, flip back
    flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=dev)
    packed = torch.bitwise_xor(keys, flip)
    reduce_first_hit(packed, group=group)
    keys = torch.bitwise_xor(packed, flip).contiguous()
    idx = torch.empty(R, dtype=torch.int32, device=dev)
    t = torch.empty(R, dtype=torch.float32, device=dev)
    _lib.call("drt_first_hit_finalize", ptr(keys), R, int(total_triangles), bs, ptr(idx), ptr(t), stream())
    return idx, t, keys
