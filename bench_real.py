"""Legs on the reference's OWN meshes: bruxelles.obj is the mesh of the reference's benchmark
harness (differt/tests/benchmarks/fixtures.py:43-68: one transmitter 10 m above the mesh centre, one receiver 10 m to
the side at 1.5 m; test_rt.py:77-196: any hit / first hit / visibility with 10 000 lattice rays on a random 50 % mask,
`trace_paths` orders 0 and 1 exhaustive and hybrid after `set_assume_quads()`), read from tests/golden/*.npz.

    python bench_real.py

Beyond the harness: orders 2 and 3 of 16 TX x 64 RX through the pruned search, with the pairing pass (coplanar pairs
found anywhere in the triangle soup) and triangle by triangle (`pairs=False`: what round 4 ran on every real mesh).
"""

from __future__ import annotations

import json
import time

import numpy as np


def _time(fn, reps=5, warm=1):
    import torch

    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def harness_legs(G, V, Tr) -> dict:
    """The reference harness's own calls, shapes and arguments (test_rt.py:77-196)."""
    import torch

    rng = np.random.default_rng(0)
    mask = rng.random(Tr.shape[0]) < 0.5  # Mesh.sample(0.5, by_masking=True)
    centre = V.mean(0)
    tx = (centre + np.array([0, 0, 10.0], np.float32)).astype(np.float32)[None]
    rx = (centre + np.array([10.0, 0, 1.5], np.float32)).astype(np.float32)[None]
    mesh = G.Mesh(V, Tr, mask=mask)
    d = G.fibonacci_lattice(10_000)
    o = torch.as_tensor(tx, device="cuda").expand(10_000, 3).contiguous()
    mesh.first_triangle_hit_by_ray(o[:8], d[:8], accel="bvh")  # builds the LBVH once (the reference's Warp mesh is cached too)
    out = {"rays": 10_000, "triangles": int(Tr.shape[0]), "active_triangles": int(mask.sum())}
    for accel in (None, "bvh"):
        tag = "bvh" if accel else "brute"
        t, hit = _time(lambda: mesh.ray_intersect_any_triangle(o, d, accel=accel), reps=20)
        out[f"any_hit_{tag}_us"] = t * 1e6
        t, (idx, _) = _time(lambda: mesh.first_triangle_hit_by_ray(o, d, accel=accel), reps=20)
        out[f"first_hit_{tag}_us"] = t * 1e6
        out[f"first_hit_{tag}_hits"] = int((idx >= 0).sum())
        t, vis = _time(lambda: mesh.triangles_visible_from_vertex(torch.as_tensor(tx, device="cuda"), num_rays=10_000, accel=accel), reps=10)
        out[f"visibility_{tag}_us"] = t * 1e6
        out[f"visible_triangles_{tag}"] = int(vis.sum())
    # compute_paths: orders 0 and 1, assume_quads as the harness sets it, exhaustive (with and without disconnecting the
    # inactive triangles) and hybrid
    qmesh = G.Mesh(V, Tr, mask=mask, assume_quads=True)
    scene = G.Scene(torch.as_tensor(tx, device="cuda"), torch.as_tensor(rx, device="cuda"), qmesh)
    for name, mk in (("exhaustive_no_disconnect", lambda: G.ExhaustivePathTracer(disconnect_inactive_triangles=False)),
                     ("exhaustive_disconnect", lambda: G.ExhaustivePathTracer(disconnect_inactive_triangles=True)),
                     ("hybrid", lambda: G.HybridPathTracer(num_rays=10_000))):
        def step():
            n = 0
            for order in (0, 1):
                n += int(scene.trace_paths(order=order, solver=mk()).num_valid_paths)
            return n

        try:
            t, n = _time(step, reps=5)
            out[f"trace_paths_orders01_{name}_ms"] = t * 1e3
            out[f"trace_paths_orders01_{name}_valid"] = n
        except Exception as exc:  # noqa: BLE001
            out[f"trace_paths_orders01_{name}_error"] = repr(exc)[:200]
    return out


def beam_legs(G, V, Tr, ntx=16, nrx=64, orders=(2, 3)) -> dict:
    import torch

    import synthetic_scenes as S

    tx, rx = S.outdoor_end_points(G, V, Tr, ntx, nrx)
    mesh = G.Mesh(V, Tr)
    info = mesh.beam_pairing()
    out = {"num_tx": ntx, "num_rx": nrx, "triangles": int(Tr.shape[0]), "pair_mode": info["pair_mode"],
           "primitives": info["primitives"], "paired_primitives": info["pairs"]}
    tracer = G.ExhaustivePathTracer(accel="bvh")
    for order in orders:
        for pairs in (True, False):
            def step():
                txg = torch.tensor(tx, device="cuda", requires_grad=True)
                scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mesh)
                p = tracer.trace_beam_pruned(scene, order, pairs=pairs, max_paths=1 << 18)
                if p.objects.shape[0]:
                    torch.sqrt((torch.diff(p.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
                return p.objects.shape[0]

            try:
                t, nv = _time(step, reps=2 if order == 3 else 5)
                st = tracer.last_beam_stats
                out[f"beam_order{order}" + ("" if pairs else "_triangles")] = {
                    "s_per_step": t, "valid_paths": int(nv), "valid_paths_per_s": nv / t, "pair_mode": bool(st["pair_mode"]),
                    "prefix_levels": st["levels"], "rows_traced": st["rows"],
                    "kernel_ms": {"last_expansion": st["expand_last_ms"], "receiver_stage": st["emit_ms"],
                                  "row_sort_and_trace": st["trace_ms"]}}
            except Exception as exc:  # noqa: BLE001
                out[f"beam_order{order}" + ("" if pairs else "_triangles")] = {"error": repr(exc)[:300]}
    # the exhaustive order-2 space of the same end points: the count the pruned legs must reproduce
    try:
        scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
        t, ex = _time(lambda: tracer.trace_rank_range_literal(scene, 2, max_survivors=1 << 25, max_paths=1 << 20), reps=1)
        n = Tr.shape[0]
        out["exhaustive_order2"] = {"s_per_step": t, "valid_paths": int(ex.objects.shape[0]),
                                    "path_candidates_per_s": ntx * nrx * n * (n - 1) / t}
        if "error" not in out.get("beam_order2", {"error": 1}):
            out["beam_order2"]["same_valid_paths_as_exhaustive"] = out["beam_order2"]["valid_paths"] == int(ex.objects.shape[0])
    except Exception as exc:  # noqa: BLE001
        out["exhaustive_order2"] = {"error": repr(exc)[:300]}
    return out


def run(names=("bruxelles", "manhattan", "manhattan_small")) -> dict:
    import differt_amd.geometry as G
    import synthetic_scenes as S

    out = {}
    for name in names:
        V, Tr = S.load_real_mesh(name)
        rec = beam_legs(G, V, Tr)
        try:  # whole order-3 pair spaces of this mesh through the exhaustive tracer: the committed record, quoted while fresh
            from bench_paths import exhaustive_record

            if isinstance(rec.get("beam_order3"), dict) and "error" not in rec["beam_order3"]:
                rec["beam_order3"]["same_valid_paths_as_exhaustive"] = exhaustive_record(f"{name} order 3")
        except Exception:  # noqa: BLE001
            pass
        if name == "bruxelles":
            rec["reference_harness"] = harness_legs(G, V, Tr)
        out[name] = rec
    return out


if __name__ == "__main__":
    print(json.dumps(run(), indent=1))
