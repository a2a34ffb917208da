"""Image-method tracing benchmark (BASELINE configs[2]): 16 TX x 64 RX, 10k-triangle synthetic
Manhattan mesh, reflection order 2, forward + gradient w.r.t. the TX positions.

Used by bench.py (the "paths" object of its JSON line) and runnable on its own:

    python bench_paths.py [--ranks N] [--order 2]

One step = candidate ranks [0, N) of the n*(n-1) order-2 candidates, unranked on the GPU, for every
(tx, rx) pair: filter kernel -> occlusion kernel -> sort -> emit, then loss = sum of valid path
lengths and its backward through the VJP kernel.  Reports path-candidates/s and valid paths/s.
"""

from __future__ import annotations

import argparse
import json
import time

import numpy as np


def run(dev=None, order: int = 2, num_ranks: int | None = None, num_tx: int = 16, num_rx: int = 64,
        num_boxes: int = 1000, steps: int = 2, cpu_sample: bool = True, rank: int = 0, world: int = 1,
        dist=None, checkpoint=None) -> dict:
    """With world > 1 the candidate-rank space is cut in one contiguous block per rank
    (differt_amd.distributed.shard_interval): no collective during compute.  Time = max over ranks,
    valid paths = sum over ranks (the gather / gradient all-reduce epilogue of
    differt_amd.distributed is exercised by tests/test_distributed_cpu.py, not timed here)."""
    import torch

    import differt_amd.geometry as G
    import synthetic_scenes as S

    V, Tr, centres, heights = S.manhattan(num_boxes)
    tx, rx = S.manhattan_tx_rx(centres, heights, num_tx, num_rx)
    mesh = G.Mesh(V, Tr)
    n = mesh.num_primitives
    total = n * (n - 1) ** (order - 1)
    count = total if num_ranks is None else min(num_ranks, total)
    # occlusion of the survivors on the mesh LBVH (bit-identical to the brute-force stage, tests/test_bvh_gpu.py;
    # 17 ms -> under 1 ms for the 8.4e5 survivors of a configs[2] step; built once per mesh, outside the timed steps)
    tracer = G.ExhaustivePathTracer(accel="bvh")
    timed = G.ExhaustivePathTracer(accel="bvh", collect_stats=True)  # one extra, untimed step for the per-kernel times
    from differt_amd.distributed import shard_interval

    lo, hi = shard_interval(count, world, rank)

    def step(tr=tracer):
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
        paths = tr.trace_rank_range_literal(scene, order, lo, hi, max_survivors=1 << 24, max_paths=1 << 20)
        loss = torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum()
        loss.backward()
        return paths.objects.shape[0], txg.grad

    # The timed region is collective-free (each rank works on its own block); ONE all-reduce at the
    # end combines {max time, sum of valid paths, sum of |grad|, failure flag}, and every rank reaches
    # it even if its own leg raised -- a failing rank cannot dead-lock the others.
    nvalid, grad, dt, err, stage = 0, None, 0.0, None, None
    try:
        nvalid, grad = step()  # warm-up (also sizes the queues)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            nvalid, grad = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        step(timed)  # HIP-event stage times (drt_trace_stats) of one more, untimed step
        stage = dict(timed.last_stats)
    except Exception as exc:  # noqa: BLE001
        err = repr(exc)
    if dist is not None and world > 1:
        gsum = float(grad.abs().sum().item()) if grad is not None else 0.0
        vmax = torch.tensor([dt, 1.0 if err else 0.0], dtype=torch.float64, device="cuda")
        vsum = torch.tensor([float(nvalid), gsum], dtype=torch.float64, device="cuda")
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
        dt, failed = float(vmax[0].item()), bool(vmax[1].item() > 0)
        nvalid = int(vsum[0].item())
        if failed and err is None:
            err = "another rank failed"
    if err is not None:
        return {"error": err}
    nvalid = int(nvalid)
    pairs = num_tx * num_rx
    out = {
        "workload": f"configs[2]: {num_tx} TX x {num_rx} RX, {Tr.shape[0]}-triangle synthetic Manhattan mesh, "
                    f"order {order}, fwd + grad(TX); candidate ranks [0, {count}) of {total} per pair"
                    + (f", rank space sharded over {world} GPUs (strong scaling)" if world > 1 else ""),
        "n_gpus": world,
        "occlusion_stage": "LBVH walk per survivor (DRT_TRACE_USE_BVH)",
        "path_candidates_per_step": pairs * count,
        "valid_paths": nvalid,
        "s_per_step": dt,
        "path_candidates_per_s": pairs * count / dt,
        "valid_paths_per_s": nvalid / dt,
        "roofline": paths_roofline(order, stage),
        "grad_tx_finite": bool(torch.isfinite(grad).all().item()),
        "grad_tx_absmax": float(grad.abs().max().item()),
    }
    import sys

    def note(msg):
        print(f"[bench_paths] {msg}", file=sys.stderr, flush=True)
        if checkpoint is not None:  # the legs so far, on disk: a later leg that dies or times out does not take them along
            checkpoint(out)

    if rank == 0 and world == 1 and num_ranks is None and order == 2:
        note("exhaustive done")
        out["beam_pruned"] = beam_leg(G, mesh, tx, rx, order, nvalid)
        note("beam_pruned done")
        out["beam_pruned_graph"] = beam_graph_leg(G, mesh, tx, rx, order, nvalid)
        note("beam_pruned_graph done")
        # BASELINE configs[3]: the same scene at order 3 -- 1.02e15 candidates, reachable only through the pruned
        # search (no exhaustive count to compare with: 91 paths, re-validated by the oracle in tests/test_full_size_gpu.py)
        out["beam_pruned_order3"] = beam_leg(G, mesh, tx, rx, 3, None, reps=1)
        out["beam_pruned_order3"]["same_valid_paths_as_exhaustive"] = exhaustive_record("configs[3]")
        # configs[3] in ONE pass and one HIP graph (29 GiB of workspace at capacities of twice the measured list sizes:
        # what 288 GB of HBM are for)
        note("beam_pruned_order3 done")
        out["beam_pruned_order3_graph"] = beam_graph_leg(G, mesh, tx, rx, 3, out["beam_pruned_order3"].get("valid_paths"), reps=2)
        note("beam_pruned_order3_graph done")
        # the same configs as QUAD meshes (assume_quads=True: the city is boxes, and the reference's own harness calls
        # set_assume_quads(), tests/benchmarks/test_rt.py:162): exhaustive order 2, pruned orders 2 and 3
        try:
            qmesh = G.Mesh(V, Tr, assume_quads=True)
            out["assume_quads"] = quads_legs(G, qmesh, tx, rx)
        except Exception as exc:  # noqa: BLE001
            out["assume_quads"] = {"error": repr(exc)}
        note("assume_quads done")
        out["visibility_pruned"] = pruned_leg(G, mesh, tx, rx, order, nvalid)
        note("visibility_pruned done")
    if rank == 0 and world == 1 and num_ranks is None and order == 2:
        # the drop-in itself: Scene.trace_paths(order, chunk_size=...) in the reference's dense layout (bench_dense.py)
        try:
            import bench_dense

            out["reference_api"] = bench_dense.legs()
        except Exception as exc:  # noqa: BLE001
            out["reference_api"] = {"error": repr(exc)}
        note("reference_api done")
    if rank == 0 and world == 1 and num_ranks is None and order == 2:
        # the reference's own meshes (bruxelles.obj = the mesh of its benchmark harness): harness shapes + pruned orders 2 / 3
        try:
            import bench_real

            out["real_meshes"] = bench_real.run()
        except Exception as exc:  # noqa: BLE001
            out["real_meshes"] = {"error": repr(exc)}
        note("real_meshes done")
    if cpu_sample and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_sample_rate(V, Tr, tx, rx, order, n)
    return out


PEAK_VALU_ISSUE = 1024 * 32 * 2.4e9  # lane-ops/s: 1024 SIMD-32, one wave-instruction per 2 cycles, 2.4 GHz
PEAK_FP32_FMA = 157.3e12            # MI355X_MICROARCH.md FP32 vector peak (counts an FMA as 2 flop)


def paths_roofline(order: int, stage: dict | None) -> dict | None:
    """Roofline of the kernel that owns the paths metric (the filter stage), from quantities that are
    MEASURED: kernel time = HIP events around the launch (drt_trace_stats, this run); executed VALU
    instructions per (tx, rx, candidate) = SQ_INSTS_VALU x 64 / candidates of the committed counter
    pass (profiles/r02/pmc_trace_filter.json, `rocprofv3 --pmc SQ_INSTS_VALU ...` on this same step).
    achieved = executed lane-operations / s; peak = the non-FMA issue ceiling (one-rounding-per-
    operation code cannot use FMA), with the fraction of the 157.3 TFLOP/s FMA peak beside it."""
    import json
    from pathlib import Path

    if not stage or not stage.get("filter_ms"):
        return None
    root = Path(__file__).resolve().parent / "profiles"
    # newest committed counter record; it carries a hash of the kernel sources it was collected on
    pmcs = sorted(root.glob("r*/pmc_trace_filter.json"))
    pmc = pmcs[-1] if pmcs else root / "r02" / "pmc_trace_filter.json"
    per_cand = None
    occ_per_surv_tri = None
    pmc_stale = None
    if pmc.exists():
        try:
            from differt_amd._srchash import is_stale

            rec = json.loads(pmc.read_text())
            pmc_stale = is_stale(rec, "trace_filter")
            per_cand = rec.get("executed_valu_per_candidate", {}).get(str(order))
            occ_per_surv_tri = rec.get("occlusion_valu_per_survivor_triangle")
        except Exception:  # noqa: BLE001
            per_cand = None
    out = {
        "kernel": f"drt::trace_filter_kernel<{order}, false>",
        "bound": "valu",
        "kernel_ms": stage["filter_ms"],
        "candidates_per_launch": stage["candidates"],
        "survivors": stage["survivors"],
        "occlusion_kernel_ms": stage["occlusion_ms"],
        "sort_emit_ms": stage["sort_emit_ms"],
        "executed_valu_per_candidate": per_cand,
        "executed_valu_source": f"profiles/{pmc.parent.name}/pmc_trace_filter.json (SQ_INSTS_VALU pass, committed)",
        "pmc_stale": pmc_stale,
        "peak": PEAK_VALU_ISSUE,
        "peak_fma_flops": PEAK_FP32_FMA,
        "unit": "lane-ops/s",
    }
    if per_cand:
        ach = per_cand * stage["candidates"] / (stage["filter_ms"] * 1e-3)
        out.update({"achieved": ach, "frac": ach / PEAK_VALU_ISSUE, "frac_of_157TF": ach / PEAK_FP32_FMA})
    if occ_per_surv_tri and stage.get("occlusion_ms"):
        out["occlusion_valu_per_survivor_triangle"] = occ_per_surv_tri
    return out


def beam_leg(G, mesh, tx, rx, order: int, expected_valid: int | None, reps: int = 3) -> dict:
    """The same step through ExhaustivePathTracer.trace_beam_pruned: FULL coverage of the candidate space
    with geometric (conservative) pruning instead of evaluating every candidate -- same valid paths, same
    order, same vertex bits as the exhaustive step."""
    import torch

    tracer = G.ExhaustivePathTracer(accel="bvh")  # occlusion of the few surviving rows on the LBVH

    def step():
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
        paths = tracer.trace_beam_pruned(scene, order)
        torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
        return paths.objects.shape[0]

    try:
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            nv = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        st = tracer.last_beam_stats
        return {"s_per_step": dt, "valid_paths": int(nv), "valid_paths_per_s": nv / dt,
                "same_valid_paths_as_exhaustive": None if expected_valid is None else int(nv) == int(expected_valid),
                "order": order, "kappa": 64.0, "assume_quads": bool(mesh.assume_quads),
                "coplanar_pair_mode": bool(st["pair_mode"]),
                "rows_traced": int(st["rows"]), "prefix_levels": st["levels"], "unit_m": st["unit_m"], "grazing_prefixes": st["grazing_prefixes"],
                "kernel_ms": {"last_expansion": st["expand_last_ms"], "receiver_stage": st["emit_ms"],
                              "row_sort_and_trace": st["trace_ms"]},
                "roofline": beam_roofline(order, st, bool(mesh.assume_quads)),
                "entry_point": "drt_trace_paths_beam (one native call per step)",
                "coverage": "all n(n-1)^(order-1) candidates of every (tx, rx) pair; error bounds per mirror from its "
                            "incidence geometry (no smallest-cosine parameter)"}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def beam_graph_leg(G, mesh, tx, rx, order: int, expected_valid: int | None, reps: int = 10, max_paths: int = 4096) -> dict:
    """The pruned step as an XLA executable would run it (reference boundary: wp.jax_callable with fixed output_dims,
    geometry/_mesh.py:266-276): `drt_trace_paths_beam_async` (static shapes, list sizes on the device, no host
    synchronisation, no allocation) + the loss cotangent + `drt_trace_paths_vjp` over all `max_paths` rows (padding rows
    carry key -1 and contribute nothing), captured ONCE in a HIP graph and replayed: forward + grad(TX) per replay,
    no Python or host round trip inside the step."""
    import ctypes as C

    import torch

    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream

    try:
        tracer = G.ExhaustivePathTracer(accel="bvh")
        txd, rxd = torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda")
        scene = G.Scene(txd, rxd, mesh)
        # static capacities as a deployment would fix them: twice what one synchronous call measured on this scene
        # (the one-pass form launches every stage over the CAPACITY of its input list; an overflow is reported in
        # counts[2], never silent)
        tracer.trace_beam_pruned(scene, order)
        st0 = tracer.last_beam_stats
        p2 = lambda v: 1 << max(int(v) - 1, 1).bit_length()  # noqa: E731
        caps = {"max_records": p2(2 * st0["levels"][-1]), "max_rows": p2(2 * st0["rows"]),
                # rows that pass the geometric checks: about one triangle row per pair row in coplanar-pair mode
                "max_survivors": p2(max(st0["rows"] // 2, 1 << 20))}
        if order >= 3:
            caps["max_entries"] = p2(2 * st0["levels"][-2])
        out = tracer.trace_beam_pruned_static(scene, order, max_paths=max_paths, **caps)
        gtx, grx = torch.zeros_like(txd), torch.zeros_like(rxd)
        gmv = torch.zeros_like(mesh.vertices)
        cands = _lib.Candidates()
        cands.table, cands.num_nodes, cands.order = None, mesh.num_primitives, order
        cands.reserved = _lib.DRT_CAND_PACKED_KEYS
        h = mesh.handle().h

        def launch():
            tracer.trace_beam_pruned_static(scene, order, max_paths=max_paths, out=out, **caps)
            v = out["vertices"]
            seg = v[:, 1:] - v[:, :-1]
            ln = torch.sqrt((seg * seg).sum(-1, keepdim=True))
            unit = torch.where(ln > 0, seg / ln, torch.zeros_like(seg))  # d(sum of segment lengths) / d(segment)
            cot = torch.zeros_like(v)
            cot[:, 1:] += unit
            cot[:, :-1] -= unit
            gtx.zero_(); grx.zero_(); gmv.zero_()
            _lib.call("drt_trace_paths_vjp", h, ptr(txd), txd.shape[0], ptr(rxd), rxd.shape[0], C.byref(cands),
                      ptr(out["keys"]), ptr(cot), max_paths, ptr(gtx), ptr(grx), ptr(gmv), stream())

        # (the first static call above is still in flight on the current stream and owns `out`: the warm-up on the side
        # stream must not start before it has finished -- two instances on one workspace corrupt each other's lists; found
        # in round 5 as an intermittent memory fault of this leg once the row sort addressed its scatter by counters)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            launch()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        c = out["counts"].tolist()
        nv = int(c[1])
        # the gradient against the synchronous autograd path
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        ref = tracer.trace_beam_pruned(G.Scene(txg, rxd, mesh), order)
        torch.sqrt((torch.diff(ref.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
        scale = float(txg.grad.abs().max()) + 1e-30
        return {"s_per_step": dt, "valid_paths": nv, "valid_paths_per_s": nv / dt, "status_word": int(c[2]),
                "same_valid_paths_as_exhaustive": None if expected_valid is None else nv == int(expected_valid),
                "same_keys_as_sync_entry": bool(torch.equal(out["keys"][:nv], ref.keys)),
                "grad_tx_max_rel_diff_vs_sync": float((gtx - txg.grad).abs().max()) / scale,
                "order": order, "max_paths": max_paths, "capacities": caps,
                "entry_point": "drt_trace_paths_beam_async + drt_trace_paths_vjp in ONE HIP graph (capture once, replay per step)"}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def beam_roofline(order: int, st: dict, quads: bool) -> dict | None:
    """VALU-issue roofline of the kernel that owns a pruned step (the last expansion, order >= 2): kernel time = HIP
    events around its launches in THIS run (drt_beam_stats.expand_last_ms); executed VALU instructions per step from the
    committed counter pass of the same step (profiles/r*/pmc_beam_expand.json, hash-stamped: `pmc_stale` when the
    kernel sources changed since).  Peak = one wave instruction per SIMD every 2 cycles (no packed / FMA credit)."""
    import json
    from pathlib import Path

    if order < 2 or not st.get("expand_last_ms"):
        return None
    recs = sorted((Path(__file__).resolve().parent / "profiles").glob("r*/pmc_beam_expand.json"))
    out = {"bound": "valu", "kernel_ms": st["expand_last_ms"], "peak": PEAK_VALU_ISSUE, "peak_fma_flops": PEAK_FP32_FMA,
           "unit": "lane-ops/s", "achieved": None, "frac": None, "frac_of_157TF": None, "pmc_stale": None}
    if not recs:
        return out
    try:
        from differt_amd._srchash import is_stale

        rec = json.loads(recs[-1].read_text())
        legs = rec.get("legs", {})
        # a quad mesh and a triangle mesh in coplanar-pair mode run the same `_q4` kernel over the same n/2 primitives:
        # the counter pass of either describes both when only one was collected
        leg = legs.get(f"order{order}{'_quads' if quads else ''}") or legs.get(f"order{order}")
        out["pmc_stale"] = is_stale(rec, "beam")
        out["source"] = f"profiles/{recs[-1].parent.name}/pmc_beam_expand.json"
        if leg:
            out["kernel"] = leg.get("kernel")
            valu = leg["SQ_INSTS_VALU_per_step"]
            out["executed_valu_wave_instructions_per_step"] = valu
            out["achieved"] = valu * 64 / (st["expand_last_ms"] * 1e-3)
            out["frac"] = out["achieved"] / PEAK_VALU_ISSUE
            out["frac_of_157TF"] = out["achieved"] / PEAK_FP32_FMA
            for k in ("SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES", "valu_issue_frac_from_counters"):
                if k in leg:
                    out[k] = leg[k]
    except Exception:  # noqa: BLE001
        pass
    return out


def quads_legs(G, qmesh, tx, rx) -> dict:
    import torch

    nq = qmesh.num_primitives
    tracer = G.ExhaustivePathTracer(accel="bvh")

    def step():
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), qmesh)
        paths = tracer.trace_rank_range_literal(scene, 2, max_survivors=1 << 24, max_paths=1 << 20)
        torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
        return paths.objects.shape[0]

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nv = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cand = len(tx) * len(rx) * nq * (nq - 1)
    out = {"workload": f"configs[2] / configs[3] scene as {nq} quads (assume_quads=True)",
           "exhaustive_order2": {"s_per_step": dt, "path_candidates_per_step": cand, "path_candidates_per_s": cand / dt,
                                 "valid_paths": int(nv)},
           "beam_pruned_order2": beam_leg(G, qmesh, tx, rx, 2, nv),
           "beam_pruned_order3": beam_leg(G, qmesh, tx, rx, 3, None, reps=1)}
    return out


def exhaustive_record(config: str) -> dict | None:
    """Order 3 has no exhaustive count for the whole 16 x 64 problem (1.02e15 candidates); what exists is the committed
    record of scratch/exhaustive_pairs.py: the WHOLE 9.998e11-candidate space of single pairs through the exhaustive
    tracer == the pruned search's rows of those pairs (objects and vertex bits, kappa = 64 and 1), plus the 1-pair
    slice of it in the -m gpu suite (tests/test_full_size_gpu.py)."""
    import json
    from pathlib import Path

    from differt_amd._srchash import source_hash

    recs = sorted((Path(__file__).resolve().parent / "profiles").glob("r*/stress/exhaustive_pairs.json"))
    if not recs:
        return None
    try:
        data = json.loads(recs[-1].read_text())
        # a record describes the kernels it was taken on: one from another tree (round 5 quoted round 4's, taken on a kernel
        # that no longer was the default mapping) is not evidence for this one
        stamp = data.get("source_hash") or {}
        if any(stamp.get(k) != source_hash(k) for k in ("beam", "trace_filter")):
            return None
        for r in data["records"]:
            if r["config"] == config:
                return {"checked_pairs": r["checked_pairs"], "all_equal": r["all_equal"],
                        "candidates_evaluated": r["candidates_evaluated"], "kappas": r["kappas"],
                        "record": f"profiles/{recs[-1].parent.parent.name}/stress/exhaustive_pairs.json"}
    except Exception:  # noqa: BLE001
        return None
    return None


def pruned_leg(G, mesh, tx, rx, order: int, expected_valid: int) -> dict:
    """Same step through HybridPathTracer.trace_rank_range: first / last interaction restricted to the
    primitives visible from the transmitters / receivers (reference _solvers.py:996-1056), the pruned
    product space unranked on the GPU.  Must find the same valid paths as the exhaustive step."""
    import torch

    solver = G.HybridPathTracer(num_rays=1_000_000, accel="bvh")

    def step():
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
        paths = solver.trace_rank_range(scene, order, max_survivors=1 << 24, max_paths=1 << 20)
        torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
        return paths.objects.shape[0], scene

    try:
        nv, scene = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nv, scene = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        first, last = (int(x.shape[0]) for x in solver._visible_sets(scene)[:2])
        out = {"s_per_step": dt, "valid_paths": int(nv), "same_valid_paths_as_exhaustive": int(nv) == int(expected_valid),
               "visible_first": first, "visible_last": last,
               "pruned_candidates_per_pair": solver.num_path_candidates(scene, order),
               "note": "includes the visibility estimation (80 viewpoints x 1e6 rays on the LBVH)"}

        # extension: pruning per (tx, rx) pair, all pairs in one ragged launch (HybridPathTracer.trace_pairs);
        # visibility = 1e5 lattice rays per viewpoint + 7 interior sample points per face
        psolver = G.HybridPathTracer(num_rays=100_000, accel="bvh", sample_triangles=True)

        def pstep():
            txg = torch.tensor(tx, device="cuda", requires_grad=True)
            scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
            paths = psolver.trace_pairs(scene, order)
            torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
            return paths.objects.shape[0]

        pstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        npp = pstep()
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        out["per_pair"] = {"s_per_step": dtp, "valid_paths": int(npp), "valid_paths_per_s": npp / dtp,
                           "same_valid_paths_as_exhaustive": int(npp) == int(expected_valid),
                           "candidate_evals_per_step": int(psolver.last_num_evaluated),
                           "visibility": "1e5 lattice rays + 7 sample points per face, per viewpoint"}
        # the same step with the visibility estimate reused (end points that move little between steps)
        vis = psolver.estimate_visibility(G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh))

        def cstep():
            txg = torch.tensor(tx, device="cuda", requires_grad=True)
            scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
            paths = psolver.trace_pairs(scene, order, visibility=vis)
            torch.sqrt((torch.diff(paths.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
            return paths.objects.shape[0]

        cstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ncp = cstep()
        torch.cuda.synchronize()
        out["per_pair"]["s_per_step_cached_visibility"] = (time.perf_counter() - t0) / 5
        out["per_pair"]["valid_paths_cached_visibility"] = int(ncp)
        return out
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def cpu_sample_rate(V, Tr, tx, rx, order, n, budget_s: float = 10.0) -> dict:
    """The CPU oracle (dense reference algorithm incl. brute-force occlusion of every candidate) on a
    bounded sample: 1 TX x 1 RX x the first `m` candidates."""
    import os

    import oracle as orc

    m = 20000
    cand = orc.CompleteGraphIter(n, n, n + 1, order + 2, False).collect_array(m).astype(np.int32)
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.trace_path_candidates(V, Tr, tx[:1], rx[:1], cand)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 500:
            break
    return {
        "value": m * reps / el,
        "unit": "path-candidates/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": f"{reps} passes of 1 TX x 1 RX x first {m} order-{order} candidates, dense reference "
                  f"algorithm (every candidate's {order + 1} segments tested against all {Tr.shape[0]} triangles), {el:.1f} s",
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--ranks", type=int, default=None)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--boxes", type=int, default=1000)
    ap.add_argument("--tx", type=int, default=16)
    ap.add_argument("--rx", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(order=a.order, num_ranks=a.ranks, steps=a.steps, num_boxes=a.boxes, num_tx=a.tx,
                         num_rx=a.rx, cpu_sample=not a.no_cpu)))
