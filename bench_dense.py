"""The drop-in itself: `Scene.trace_paths(order, chunk_size=...)` iterated to exhaustion -- the reference's own call
(reference geometry/_scene.py:735-764 -> _solvers.py:850-957 -> 499-770), dense `[Ntx,Nrx,C,...]` outputs for EVERY
candidate, as `differt/tests/benchmarks/test_rt.py:151-196` drives it (that harness also calls `set_assume_quads()`).

    python bench_dense.py [--order 2] [--rx 64] [--chunk 1048576] [--quads] [--max-chunks N]

Workload: a sub-block of BASELINE configs[2] -- 1 TX x 64 RX on the 10 000-triangle synthetic Manhattan mesh, all
99 990 000 order-2 candidates (6.4e9 (tx, rx, candidate) rows), in chunks of 2^20 candidates; per chunk the solver
fills the candidate table on the GPU (`drt_candidates_fill`), then `drt_trace_paths_dense_ex` writes vertices,
objects, mask and interaction types for every row.  The operator is HBM-write bound: SURVEY.md section 8d prices it at
81 B per row at order 2 (48 vertices + 16 objects + 1 mask + 8 types written, 8 candidate ids read), 105 B at order 3.

Numbers reported:
  * `candidates_per_s`            whole loop, wall clock between two synchronisations (Python, the fill kernel and the
                                  `num_valid_paths` of the consumer included) -- what a DiffeRT user sees;
  * `roofline`                    the dense kernel alone: HIP events around the launch on its stream (drt_trace_stats,
                                  taken on a second, untimed pass), `achieved = 81 B x rows / kernel time`,
                                  `frac = achieved / 8 TB/s`; `written_frac` counts only the 73 B a row really
                                  writes (the candidate ids are read once per chunk, not once per row).
"""

from __future__ import annotations

import argparse
import json
import time

HBM_PEAK_BPS = 8.0e12  # MI355X_MICROARCH.md


def bytes_per_row(order: int) -> tuple[int, int]:
    """(SURVEY 8d algorithmic bytes per (tx, rx, candidate), bytes actually written per row)."""
    written = 12 * (order + 2) + 4 * (order + 2) + 1 + 4 * order
    return written + 4 * order, written


def fill_probe() -> float:
    """Write bandwidth of THIS box right now: a 4 GiB `fill_` (torch's vectorised store kernel), GB/s.  Boxes of the pool
    differ by ~20 % on pure-store kernels; the probe says what the dense tracer's rate has to be read against."""
    import torch

    t = torch.empty(1 << 32, dtype=torch.uint8, device="cuda")
    t.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        t.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    return 5 * (1 << 32) / (e0.elapsed_time(e1) * 1e-3) / 1e9


def run(order: int = 2, num_tx: int = 1, num_rx: int = 64, chunk: int = 1 << 20, assume_quads: bool = False,
        max_chunks: int | None = None, num_boxes: int = 1000, stat_chunks: int = 8) -> dict:
    import torch

    import differt_amd.geometry as G
    import synthetic_scenes as S

    V, Tr, centres, heights = S.manhattan(num_boxes)
    tx, rx = S.manhattan_tx_rx(centres, heights, 16, 64)
    tx, rx = tx[:num_tx], rx[:num_rx]
    mesh = G.Mesh(V, Tr, assume_quads=assume_quads)
    scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), mesh)
    n = mesh.num_primitives
    total = n * (n - 1) ** (order - 1)
    nchunks_all = -(-total // chunk)
    nchunks = nchunks_all if max_chunks is None else min(max_chunks, nchunks_all)

    def sweep(solver, limit):
        nvalid = torch.zeros((), dtype=torch.int64, device="cuda")
        rows = 0
        kernel_ms, occ_ms = [], []
        for i, paths in enumerate(scene.trace_paths(order=order, solver=solver)):
            if i >= limit:
                break
            nvalid += paths.num_valid_paths  # as the reference harness does (tests/benchmarks/test_rt.py:187-194)
            rows += paths.mask.numel()
            if solver.collect_stats:
                kernel_ms.append(solver.last_stats["filter_ms"])
                occ_ms.append(solver.last_stats["occlusion_ms"])
        return nvalid, rows, kernel_ms, occ_ms

    torch.cuda.empty_cache()  # start from a clean caching allocator: each chunk allocates ~5.5 GB of outputs
    solver = G.ExhaustivePathTracer(chunk_size=chunk)
    sweep(solver, min(3, nchunks))  # warm-up: allocator, clocks
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nvalid, rows, _, _ = sweep(solver, nchunks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nvalid = int(nvalid.item())

    timed = G.ExhaustivePathTracer(chunk_size=chunk, collect_stats=True)
    _, srows, kms, oms = sweep(timed, min(stat_chunks, nchunks))
    algo, written = bytes_per_row(order)
    # full-size chunks only (the last chunk of a sweep may be short)
    full_rows = num_tx * num_rx * min(chunk, total)
    full = [k for k in kms[: max(1, len(kms) - (1 if nchunks == nchunks_all and total % chunk else 0))]]
    kernel_ms = sum(full) / len(full)
    kernel_ms_all = [round(k, 4) for k in full]
    if kernel_ms <= 0:  # scratch A/B builds (DRT_DENSE_LEGACY) do not fill the stats: wall-clock numbers only
        return {"rows": rows, "valid_paths": nvalid, "s_total": dt, "candidates_per_s": rows / dt,
                "end_to_end_GBps": algo * rows / dt / 1e9, "roofline": None}
    achieved = algo * full_rows / (kernel_ms * 1e-3)
    out = {
        "workload": f"sub-block of configs[2]: {num_tx} TX x {num_rx} RX, {Tr.shape[0]}-triangle synthetic Manhattan mesh"
                    f"{' as ' + str(n) + ' quads (assume_quads, as the reference harness)' if assume_quads else ''}, order {order}, "
                    f"Scene.trace_paths(order, chunk_size={chunk}) iterated over {nchunks} of {nchunks_all} chunks "
                    f"({total} candidates per pair), dense [Ntx,Nrx,C,...] outputs incl. interaction types",
        "reference_call": "differt/src/differt/geometry/_scene.py:735-764 -> _solvers.py:850-957, 499-770; harness tests/benchmarks/test_rt.py:151-196",
        "rows": rows,
        "valid_paths": nvalid,
        "s_total": dt,
        "candidates_per_s": rows / dt,
        "end_to_end_GBps": algo * rows / dt / 1e9,
        "roofline": {
            "kernel": f"drt::trace_dense_kernel<{order}, {'true' if assume_quads else 'false'}>",
            "bound": "hbm",
            "bytes_per_candidate": algo,
            "bytes_written_per_candidate": written,
            "rows_per_launch": full_rows,
            "kernel_ms": kernel_ms,
            "kernel_ms_per_launch": kernel_ms_all,
            "occlusion_kernel_ms": sum(oms) / len(oms),
            "achieved": achieved / 1e9,
            "peak": HBM_PEAK_BPS / 1e9,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_BPS,
            "written_frac": written * full_rows / (kernel_ms * 1e-3) / HBM_PEAK_BPS,
            "kernel_time_source": "HIP events around the launch on its stream (drt_trace_stats), mean over "
                                  f"{len(full)} full-size launches of an untimed second pass",
            "traffic": pmc_traffic(order),
            "box_fill_GBps": fill_probe(),
        },
    }
    return out


def pmc_traffic(order: int):
    """HBM bytes per launch of the dense kernel from the committed counter pass (rocprofv3 --pmc cannot run inside
    this process): profiles/r04/pmc_trace_dense.json, hash-stamped like the other records."""
    from pathlib import Path

    root = Path(__file__).resolve().parent / "profiles"
    recs = sorted(root.glob("r*/pmc_trace_dense.json"))
    if not recs:
        return None
    try:
        from differt_amd._srchash import is_stale

        rec = json.loads(recs[-1].read_text())
        return {"bytes_per_launch": rec.get("bytes_per_launch", {}).get(str(order)),
                "source": f"profiles/{recs[-1].parent.name}/pmc_trace_dense.json",
                "pmc_stale": is_stale(rec, "trace_dense")}
    except Exception:  # noqa: BLE001
        return None


def image_method_legs() -> dict:
    """The stand-alone `image_method` (reference _solver_image_method.py:206-363) at the reference harness' shape
    (tests/benchmarks/test_rt.py:35-54: from / to of shape [3], 10 000 x 8 mirrors) and at 1e7 x 2 dense rows.
    Bytes per element: 24 (from, to) + 36 K (mirror points, normals, path) -- HBM bound."""
    import numpy as np
    import torch

    import differt_amd._lib as lib
    import differt_amd.geometry as G
    from differt_amd._tensors import ptr, stream

    rng = np.random.default_rng(1234)
    out = {}

    def unit(x):
        return x / np.linalg.norm(x, axis=-1, keepdims=True)

    def time_us(fn, n):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    for name, B, k, shared in (("harness_10000x8", 10_000, 8, True), ("dense_1e7x2", 10_000_000, 2, False),
                               ("dense_2e6x8", 2_000_000, 8, False)):
        a = torch.tensor(rng.uniform(size=(3,) if shared else (B, 3)).astype(np.float32), device="cuda")
        b = torch.tensor(rng.uniform(size=(3,) if shared else (B, 3)).astype(np.float32), device="cuda")
        mv = torch.tensor(rng.uniform(size=(B, k, 3)).astype(np.float32), device="cuda")
        mn = torch.tensor(unit(rng.normal(size=(B, k, 3))).astype(np.float32), device="cuda")
        res = torch.empty((B, k, 3), dtype=torch.float32, device="cuda")
        sa = 0 if shared else 3

        def raw():
            lib.call("drt_image_method_strided", ptr(a), sa, ptr(b), sa, ptr(mv), 3 * k, ptr(mn), 3 * k, B, k, ptr(res),
                     stream())

        n = 200 if B <= 100_000 else 30
        us_api = time_us(lambda: G.image_method(a, b, mv, mn), n)
        us_raw = time_us(raw, n)
        us_graph = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    raw()
            us_graph = time_us(g.replay, 10) / 20
        except Exception:  # noqa: BLE001
            us_graph = None
        nbytes = B * 36 * k + (24 if shared else 24 * B)
        best = us_graph or us_raw
        out[name] = {"batch": B, "mirrors": k, "from_to": "shape [3], read in place" if shared else "dense [B, 3]",
                     "us_per_call_python_api": us_api, "us_per_launch_c_abi": us_raw, "us_per_launch_hipgraph": us_graph,
                     "elements_per_s": B / (best * 1e-6), "bytes_per_launch": nbytes,
                     "hbm_frac": nbytes / (best * 1e-6) / HBM_PEAK_BPS}
        del a, b, mv, mn, res
    return out


def legs() -> dict:
    """The legs bench.py reports under paths.reference_api."""
    out = {}
    for name, kw in (("order2", {}),
                     ("order3_window", {"order": 3, "max_chunks": 24}),
                     ("order2_assume_quads", {"assume_quads": True}),
                     ("order3_assume_quads_window", {"order": 3, "assume_quads": True, "max_chunks": 24})):
        try:
            out[name] = run(**kw)
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": repr(exc)}
    if isinstance(out.get("order2"), dict) and "roofline" in out["order2"]:
        out["roofline"] = out["order2"]["roofline"]
    try:
        out["image_method"] = image_method_legs()
    except Exception as exc:  # noqa: BLE001
        out["image_method"] = {"error": repr(exc)}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--tx", type=int, default=1)
    ap.add_argument("--rx", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=1 << 20)
    ap.add_argument("--quads", action="store_true")
    ap.add_argument("--max-chunks", type=int, default=None)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--image-method", action="store_true")
    a = ap.parse_args()
    if a.image_method:
        print(json.dumps(image_method_legs()))
    elif a.all:
        print(json.dumps(legs()))
    else:
        print(json.dumps(run(order=a.order, num_tx=a.tx, num_rx=a.rx, chunk=a.chunk, assume_quads=a.quads,
                             max_chunks=a.max_chunks)))
