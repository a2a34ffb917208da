"""Strong-scaling legs of bench.py on BASELINE configs[4] (1 TX x 1024 RX, 200 000-triangle city,
order 2, forward + gradient): total work is FIXED, the N ranks split it.

* ``candidate_sharded``: the mesh (10.4 MB) is replicated, the lexicographic candidate-rank window
  ``[0, W)`` of every (tx, rx) pair is cut into one contiguous block per rank
  (differt_amd.distributed.trace_rank_range_sharded); no collective during compute.  The epilogue --
  all-gather of the valid paths + key sort (= single-GPU ``masked_vertices`` order on every rank) and
  the SUM all-reduce of the transmitter gradient -- is INSIDE the timed region.  The full space is
  4.1e13 candidates (113 s on one GPU); the default window W is sized for ~10 s at N = 1 and stated
  in the JSON.
* ``triangle_block``: the capacity path of SURVEY.md section 8e -- every rank holds T/N triangles,
  all ranks see the same rays; per batch of 2^17 rays: local packed (t, tie) keys
  (drt_first_hit_keys) -> ONE MIN all-reduce of 8 B per ray (RCCL) -> decode (drt_first_hit_finalize).
  The all-reduce of batch i is issued asynchronously and overlaps the key kernel of batch i + 1.

Both report whole-job rates with ``"scaling": "strong"``; time = max over ranks between barriers.
"""

from __future__ import annotations

import time

import numpy as np


def run(dev, rank: int = 0, world: int = 1, dist=None, *, boxes: int = 20000, rx_side: int = 32,
        window: int | None = None, steps: int = 1, tb_rays: int = 1 << 17, tb_batches: int = 8,
        tb_steps: int = 3, beam_steps: int = 3) -> dict:
    import torch

    import differt_amd._lib as lib
    import differt_amd.geometry as G
    import synthetic_scenes as S
    from differt_amd._tensors import F32_EPS, ptr, stream
    from differt_amd.distributed import (allreduce_grads, gather_paths, shard_interval,
                                         trace_rank_range_sharded)

    def note(msg):
        import sys

        print(f"[bench_scaling] rank {rank}: {msg}", file=sys.stderr, flush=True)

    def barrier():
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if dist is None or world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    V, Tr, tx, rx = S.cfg5_scene(boxes, rx_side)
    mesh = G.Mesh(V, Tr)
    n = mesh.num_primitives
    total = n * (n - 1)
    # ~10 s of work at N = 1 (3.3e11 path-candidates/s measured on one MI355X)
    W = min(total, int(window) if window else max(int(3.4e12 // max(rx.shape[0], 1)), 1))
    out: dict = {"workload": f"BASELINE configs[4]: 1 TX x {rx.shape[0]} RX, {Tr.shape[0]}-triangle synthetic Manhattan "
                             f"city, order 2, fwd + grad(TX)", "n_gpus": world, "scaling": "strong"}
    if dist is not None and world > 1:
        out["collective_backend"] = dist.get_backend()
        out["ranks_seen_by_backend"] = dist.get_world_size()

    # ---------------------------------------------------------------- candidate-rank sharding ----
    tracer = G.ExhaustivePathTracer(accel="bvh")  # survivors' occlusion on the LBVH (200k triangles)
    rx_d = torch.tensor(rx, device=dev)

    def step():
        txg = torch.tensor(tx, device=dev, requires_grad=True)
        scene = G.Scene(txg, rx_d, mesh)
        keys, verts, objs = trace_rank_range_sharded(tracer, scene, 2, 0, W, gather=False,
                                                     max_survivors=1 << 22, max_paths=1 << 16)
        loss = torch.sqrt((torch.diff(verts, dim=-2) ** 2).sum(-1)).sum()
        if verts.requires_grad:
            loss.backward()
        grad = txg.grad if txg.grad is not None else torch.zeros_like(txg)
        # epilogue (timed): every rank ends up with all valid paths in single-GPU order + the full gradient
        gk, gv, go = gather_paths(keys, verts.detach(), objs)
        allreduce_grads(grad)
        return gk, gv, go, grad

    err = None
    try:
        step()  # warm-up: sizes the queues, loads the modules, builds the mesh handle
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            gk, gv, go, grad = step()
        barrier()
        dt = max_over_ranks((time.perf_counter() - t0) / steps)
        a, b = shard_interval(W, world, rank)
        out["candidate_sharded"] = {
            "window": f"candidate ranks [0, {W}) of {total} per pair ({W / total:.3%} of the full space), "
                      f"one contiguous block of {b - a} ranks per GPU",
            "path_candidates_per_step": int(rx.shape[0]) * W,
            "s_per_step": dt,
            "path_candidates_per_s": int(rx.shape[0]) * W / dt,
            "valid_paths": int(gk.shape[0]),
            "valid_paths_per_s": int(gk.shape[0]) / dt,
            "epilogue_in_timed_region": "all_gather(counts) + all_gather(padded keys/vertices/objects) + key sort; "
                                        "SUM all-reduce of grad(TX)",
            "grad_tx_finite": bool(torch.isfinite(grad).all().item()),
            "grad_tx_absmax": float(grad.abs().max().item()),
            "keys_sorted": bool((gk[1:] > gk[:-1]).all().item()) if gk.shape[0] > 1 else True,
        }
    except Exception as exc:  # noqa: BLE001 - the headline line must still be printed
        err = repr(exc)
        out["candidate_sharded"] = {"error": err}
    note("candidate_sharded done")

    # ---------------------------------------------------------------- triangle-block sharding ----
    try:
        T = int(Tr.shape[0])
        lo, hi = shard_interval(T, world, rank)
        tv_block = mesh.triangle_vertices.detach()[lo:hi].contiguous()
        rng = np.random.default_rng(2024)  # same rays on every rank
        R = int(tb_rays)
        o_h = np.tile(tx, (R, 1)).astype(np.float32) + rng.normal(size=(R, 3)).astype(np.float32)
        d_h = rng.normal(size=(tb_batches, R, 3)).astype(np.float32)
        d_h[..., 2] = -np.abs(d_h[..., 2]) - 0.05  # shot downwards into the city
        o = torch.as_tensor(o_h, device=dev)
        d = torch.as_tensor(d_h, device=dev)
        keys = [torch.empty(R, dtype=torch.int64, device=dev) for _ in range(tb_batches)]
        idx = torch.empty((tb_batches, R), dtype=torch.int32, device=dev)
        tt = torch.empty((tb_batches, R), dtype=torch.float32, device=dev)
        flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=dev)
        eps = 10.0 * F32_EPS
        multi = dist is not None and world > 1

        def finalize(i):
            if multi:
                keys[i].bitwise_xor_(flip)
            lib.call("drt_first_hit_finalize", ptr(keys[i]), R, T, 512, ptr(idx[i]), ptr(tt[i]), stream())

        def tb_step():
            pending = None
            for i in range(tb_batches):
                lib.call("drt_first_hit_keys", ptr(o), ptr(d[i]), R, ptr(tv_block), hi - lo, lo, T, None, eps, 512,
                         ptr(keys[i]), 1, stream())
                work = None
                if multi:
                    # unsigned MIN as a signed MIN: flip the sign bit, reduce, flip back in finalize()
                    keys[i].bitwise_xor_(flip)
                    work = dist.all_reduce(keys[i], op=dist.ReduceOp.MIN, async_op=True)
                if pending is not None:  # decode batch i-1 while batch i's all-reduce is in flight
                    if pending[1] is not None:
                        pending[1].wait()
                    finalize(pending[0])
                pending = (i, work)
            if pending[1] is not None:
                pending[1].wait()
            finalize(pending[0])

        tb_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(tb_steps):
            tb_step()
        barrier()
        dt = max_over_ranks((time.perf_counter() - t0) / tb_steps)
        rays = R * tb_batches
        out["triangle_block"] = {
            "triangles_per_gpu": hi - lo,
            "rays_per_step": rays,
            "batches": f"{tb_batches} x {R} rays, all-reduce of batch i overlapped with the key kernel of batch i+1",
            "bytes_all_reduced_per_step": 8 * rays if multi else 0,
            "s_per_step": dt,
            "rays_per_s": rays / dt,
            "ray_triangle_tests_per_s": rays * T / dt,
            "hit_fraction": float((idx >= 0).float().mean().item()),
            "checksum_idx": int(idx.to(torch.int64).sum().item()),  # identical for every N (bit-exact reduce)
        }
    except Exception as exc:  # noqa: BLE001
        out["triangle_block"] = {"error": repr(exc)}
    note("triangle_block done")

    # ------------------------------------------------- full coverage: beam pruning, prefix-sharded ----
    # The COMPLETE configs[4] problem (every candidate of every pair, guarantee of DESIGN.md section 9), split
    # by (transmitter, first mirror) prefix; no collective during compute, the same epilogue as above.
    try:
        from differt_amd.distributed import trace_beam_pruned_sharded

        btracer = G.ExhaustivePathTracer(accel="bvh")

        def beam_step():
            txg = torch.tensor(tx, device=dev, requires_grad=True)
            scene = G.Scene(txg, rx_d, mesh)
            keys, verts, objs = trace_beam_pruned_sharded(btracer, scene, 2, gather=False)
            if verts.requires_grad and verts.shape[0]:
                torch.sqrt((torch.diff(verts, dim=-2) ** 2).sum(-1)).sum().backward()
            grad = txg.grad if txg.grad is not None else torch.zeros_like(txg)
            gk, gv, go = gather_paths(keys, verts.detach(), objs)
            allreduce_grads(grad)
            return gk, gv, go, grad

        beam_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(max(int(beam_steps), 1)):
            gk, gv, go, grad = beam_step()
        barrier()
        dt = max_over_ranks((time.perf_counter() - t0) / max(int(beam_steps), 1))
        st = getattr(btracer, "last_beam_stats", {})
        out["beam_sharded"] = {
            "coverage": f"all {total} candidates of each of the {rx.shape[0]} pairs ({int(rx.shape[0]) * total:.3e} "
                        f"path candidates), level-1 prefix (t, m) on rank (t n + m) mod N",
            "s_per_step": dt,
            "valid_paths": int(gk.shape[0]),
            "valid_paths_per_s": int(gk.shape[0]) / dt,
            "equivalent_path_candidates_per_s": int(rx.shape[0]) * total / dt,
            "rows_traced_this_rank": int(st.get("rows", 0)),
            "prefix_levels_this_rank": [int(x) for x in st.get("levels", [])],
            "unit_m": float(st.get("unit_m", 0.0)), "grazing_prefixes": int(st.get("grazing_prefixes", 0)),
            "checksum_keys": int(gk.sum().item()) if gk.shape[0] else 0,  # identical for every N
            "grad_tx_finite": bool(torch.isfinite(grad).all().item()),
            "grad_tx_absmax": float(grad.abs().max().item()),
            "epilogue_in_timed_region": "all_gather(counts) + all_gather(padded keys/vertices/objects) + key sort; "
                                        "SUM all-reduce of grad(TX)",
        }
    except Exception as exc:  # noqa: BLE001
        out["beam_sharded"] = {"error": repr(exc)}
    note("beam_sharded done")
    if world == 1:
        # the same complete problem as ONE HIP graph (static shapes, no host synchronisation): bench_paths.beam_graph_leg
        try:
            from bench_paths import beam_graph_leg

            out["beam_graph"] = beam_graph_leg(G, mesh, tx, rx, 2, out.get("beam_sharded", {}).get("valid_paths"))
        except Exception as exc:  # noqa: BLE001
            out["beam_graph"] = {"error": repr(exc)}
    return out


def emulate_shards(nshards: int = 8, *, window: int | None = None, reps: int = 2) -> dict:
    """Shard imbalance of the two multi-GPU splits, measured on ONE GPU: every shard of `beam_sharded` (level-1
    prefix (t, m) on rank (t n + m) mod N) and of `candidate_sharded` (one contiguous block of the rank window per
    rank) for configs[3] (order 3) and configs[4] (order 2) is run by itself, one after the other, and timed.  The
    slowest shard bounds the N-GPU step: `speedup_bound` = (unsharded step) / (slowest shard) -- what N GPUs can reach
    at best before the epilogue (two all-gathers of a few KB, one all-reduce of 48 B).  Target: max / mean <= 1.15."""
    import torch

    import differt_amd.geometry as G
    import synthetic_scenes as S
    from differt_amd.distributed import shard_interval

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            nv = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, nv

    def summary(ts, full):
        ts = [float(t) for t in ts]
        mean = sum(ts) / len(ts)
        return {"per_shard_s": ts, "max_over_mean": max(ts) / mean, "unsharded_s": full,
                "speedup_bound": full / max(ts), "sum_over_unsharded": sum(ts) / full}

    out = {"shards": nshards, "note": "every shard run alone on one MI355X, sequentially; no collective is timed"}
    for name, order in (("configs[3]", 3), ("configs[4]", 2)):
        if name == "configs[3]":
            V, Tr, c, h = S.manhattan(1000)
            tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
        else:
            V, Tr, tx, rx = S.cfg5_scene()
        mesh = G.Mesh(V, Tr)
        scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
        tracer = G.ExhaustivePathTracer(accel="bvh")
        full, nv = timed(lambda: tracer.trace_beam_pruned(scene, order).objects.shape[0])
        ts, found = [], 0
        for r in range(nshards):
            t, k = timed(lambda r=r: tracer.trace_beam_pruned(scene, order, prefix_shard=(r, nshards)).objects.shape[0])
            ts.append(t)
            found += k
        rec = {"beam_sharded": {**summary(ts, full), "valid_paths": int(nv), "valid_paths_over_shards": int(found)}}
        if order == 2:
            n = mesh.num_primitives
            total = n * (n - 1)
            W = min(total, int(window) if window else max(int(3.4e12 // max(rx.shape[0], 1)), 1))
            ex = G.ExhaustivePathTracer()
            fullc, _ = timed(lambda: ex.trace_rank_range_literal(scene, 2, 0, W, max_survivors=1 << 22).objects.shape[0])
            tc = []
            for r in range(nshards):
                lo, hi = shard_interval(W, nshards, r)
                t, _ = timed(lambda lo=lo, hi=hi: ex.trace_rank_range_literal(scene, 2, lo, hi, max_survivors=1 << 22).objects.shape[0])
                tc.append(t)
            rec["candidate_sharded"] = {**summary(tc, fullc), "window": W}
        out[name] = rec
    return out


if __name__ == "__main__":
    import argparse
    import json

    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=20000)
    ap.add_argument("--window", type=int, default=None)
    ap.add_argument("--emulate-shards", type=int, default=0,
                    help="run every shard of the beam / candidate splits alone on this GPU and report the imbalance")
    a = ap.parse_args()
    if a.emulate_shards:
        print(json.dumps(emulate_shards(a.emulate_shards, window=a.window)))
    else:
        print(json.dumps(run(torch.device("cuda", 0), boxes=a.boxes, window=a.window)))
