"""world_size-2 `gloo` tests of the multi-GPU path (candidate-rank sharding + gather, gradient
all-reduce, packed first-hit MIN reduce).  Compute is played by the CPU oracle here (the HIP path
needs a GPU); what is under test is the sharding / collective logic of differt_amd.distributed."""

from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as orc
from differt_amd.distributed import (
    allreduce_grads,
    gather_paths,
    globalize_keys,
    reduce_any_hit,
    reduce_first_hit,
    shard_interval,
)


def test_shard_interval_partitions_exactly():
    for total in (0, 1, 7, 100, 99_990_000):
        for world in (1, 2, 3, 8):
            blocks = [shard_interval(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks[:-1], blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, tmp: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from pathlib import Path

        golden = Path(__file__).resolve().parent / "golden"
        tb = json.loads((golden / "two_buildings.json").read_text())
        g = json.loads((golden / "reference_goldens.json").read_text())["advanced_path_tracing_example"]
        V, Tr = np.asarray(tb["vertices"], np.float32), np.asarray(tb["triangles"], np.int32)
        tx = np.asarray([g["tx"], [1.0, 5.0, 21.0]], np.float32)
        rx = np.asarray([g["rx"]], np.float32)
        order = 2
        cand = orc.generate_all_path_candidates(24, order).astype(np.int32)
        C = cand.shape[0]
        lo, hi = shard_interval(C, world, rank)
        # every rank traces only its block of candidate ranks, for all (tx, rx) pairs
        o = orc.trace_path_candidates(V, Tr, tx, rx, cand[lo:hi])
        P = tx.shape[0] * rx.shape[0]
        m = o["mask"].reshape(P, hi - lo)
        pair, row = np.nonzero(m)
        # what trace_rank_range returns: keys local to the window; globalize -> pair * C + rank
        local_keys = torch.tensor(pair * (hi - lo) + row, dtype=torch.int64)
        keys = globalize_keys(local_keys, hi - lo, lo, C)
        assert keys.numpy().tolist() == (pair * C + lo + row).tolist()
        verts = torch.tensor(o["vertices"].reshape(P, hi - lo, order + 2, 3)[pair, row])
        objs = torch.tensor(o["objects"].reshape(P, hi - lo, order + 2)[pair, row])
        gk, gv, go = gather_paths(keys, verts, objs)

        full = orc.trace_path_candidates(V, Tr, tx, rx, cand)
        fm = full["mask"].reshape(-1)
        assert gk.numpy().tolist() == np.flatnonzero(fm).tolist()
        assert np.array_equal(gv.numpy(), full["vertices"].reshape(-1, order + 2, 3)[fm])
        assert np.array_equal(go.numpy(), full["objects"].reshape(-1, order + 2)[fm])
        assert fm.sum() >= 2

        # gradients: SUM all-reduce of per-rank partial sums
        ga = torch.full((2, 3), float(rank + 1))
        gb = torch.full((1, 3), 10.0 * (rank + 1))
        allreduce_grads(ga, gb)
        tot = world * (world + 1) / 2
        assert torch.equal(ga, torch.full((2, 3), tot)) and torch.equal(gb, torch.full((1, 3), 10 * tot))

        # triangle-block sharding: packed (t, tie) keys, MIN over ranks == unsharded first hit
        rng = np.random.default_rng(3)
        T, R, bs = 300, 64, 11
        tvs = rng.normal(size=(T, 3, 3)).astype(np.float32)
        ro = rng.normal(size=(R, 3)).astype(np.float32)
        rd = rng.normal(size=(R, 3)).astype(np.float32) * 3
        tvs[[5, 120, 250]] = tvs[40]  # exact ties across blocks
        tlo, thi = shard_interval(T, world, rank)
        t, hit = orc.ray_intersect_triangle_dense(ro, rd, tvs[tlo:thi])
        nb = T // bs
        ntiles = nb + (1 if T % bs else 0)
        j = np.arange(tlo, thi)
        tile = np.where(j < nb * bs, j // bs, nb)
        tie = (ntiles - 1 - tile) * bs + (j - tile * bs)
        bits = t.view(np.uint32).astype(np.uint64)  # hits have t > 0: IEEE bits are order-preserving
        key = np.where(hit, (bits << np.uint64(32)) | tie[None, :].astype(np.uint64), np.uint64(2**64 - 1))
        local = key.min(axis=1)
        packed = torch.tensor((local ^ np.uint64(1 << 63)).view(np.int64))
        reduce_first_hit(packed)
        glob = packed.numpy().view(np.uint64) ^ np.uint64(1 << 63)
        ei, et = orc.first_triangle_hit_by_ray(ro, rd, tvs, batch_size=bs)
        miss = glob == np.uint64(2**64 - 1)
        gtie = (glob & np.uint64(0xFFFFFFFF)).astype(np.int64)
        gtile = ntiles - 1 - gtie // bs
        gidx = np.where(miss, -1, gtile * bs + gtie % bs)
        gt = np.where(miss, np.inf, (glob >> np.uint64(32)).astype(np.uint32).view(np.float32))
        assert np.array_equal(gidx, ei) and np.array_equal(gt.astype(np.float32), et)
        assert (ei >= 0).sum() > 10

        # triangle-block sharding of the tracer's occlusion stage: a segment is blocked if ANY rank's block
        # blocks it -> MAX all-reduce of one byte per path == the unsharded any-hit
        tol = 100 * np.finfo(np.float32).eps
        blocked_local = orc.ray_intersect_any_triangle(ro, rd, tvs[tlo:thi], hit_tol=tol)
        flags = torch.tensor(blocked_local.astype(np.uint8))
        reduce_any_hit(flags)
        assert np.array_equal(flags.numpy().astype(bool), orc.ray_intersect_any_triangle(ro, rd, tvs, hit_tol=tol))
        assert 0 < int(flags.sum()) < R
        Path(tmp, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
