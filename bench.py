#!/usr/bin/env python3
"""Headline benchmark of the MI355X DiffeRT hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): ray-triangle tests/s of `ray_intersect_triangle` (dense Moller-Trumbore,
reference geometry/_utils.py:1157-1322) on BASELINE configs[1]'s scene (10k random triangles),
fwd-only.  One *step* = one pass of the dense operator over one batch of rays that is already
resident in HBM: `--rays` rays x 10 000 triangles -> t f32[R,T] + hit u8[R,T].

configs[1] quotes 256 rays; 2.56e6 tests finish in a few microseconds (launch-bound), so the timed
batch extends the ray axis to 65 536 rays of the same seeded distribution;
the literal 256-ray launch is timed too and reported under "cfg2_literal".

With N > 1 (launched by torch.distributed.run, one rank per GPU) the ray axis is sharded: every
rank traces its own `--rays` rays against a replicated triangle set -- no data-path collective,
"scaling": "weak".  Time = max over ranks between two barriers.

The ONE JSON line stays under 4 000 characters (the driver keeps ~8 KB of stdout): the contract keys,
"roofline" (dominant kernel vs the HBM roofline, kernel time from device events on the launch
stream), "cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample, rank 0 at
N=1 only), "paths_metric" (the second half of BASELINE.json's metric: valid order-2 paths/s, fwd +
grad, configs[2]) and a flat numeric "paths" summary.  Every leg in full (image-method trace, real
meshes, strong-scaling legs, queries) goes to the sidecar named by "full" (default
gpurun_out/bench_full.json, `--full-json PATH` to move it).  On one GPU those legs run in a child process
(`--legs-in-process` keeps them here): a leg that takes its process down costs its own numbers ("legs_error"), not the line.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_ISSUE_PEAK = 256 * 4 * 32 * 2.4e9  # non-FMA FP32 lane-operations/s: 7.86e13 (the 157.3 TFLOP/s peak counts an FMA as two)
VALU_PER_TEST = 58.5  # executed one-rounding VALU instructions per ray-triangle test of mt_dense_aligned_kernel


def make_cfg2(num_rays: int, num_triangles: int, seed: int = 1234):
    """origins U([-1,1]^3)*50, directions = second point - origin
    (un-normalised segments); triangles = centre U*50 + two edge vectors N(0,1)*2."""
    rng = np.random.default_rng(seed)
    o = (rng.uniform(-1, 1, (num_rays, 3)) * 50).astype(np.float32)
    d = (rng.uniform(-1, 1, (num_rays, 3)) * 50).astype(np.float32) - o
    c = (rng.uniform(-1, 1, (num_triangles, 1, 3)) * 50).astype(np.float32)
    e = (rng.normal(size=(num_triangles, 2, 3)) * 2).astype(np.float32)
    tv = np.concatenate([c, c + e[:, :1], c + e[:, 1:]], axis=1).astype(np.float32)
    return o, d, tv


def reference_probe(o, d, tv) -> dict:
    """SURVEY 8d / BASELINE.md section 4: "if (and only if) `import jax, differt` happens to succeed on the box, the harness
    additionally times the installed package through public API calls" -- absence is reported, not hidden.  Nothing of the
    reference travels with this repository; this only looks at what the box itself has installed."""
    import importlib.util

    have = {m: importlib.util.find_spec(m) is not None for m in ("jax", "differt")}
    out = {"reference_installed": all(have.values()), "reference_probe": have}
    if not out["reference_installed"]:
        return out
    try:  # (never reached in this image: no jax, Python 3.10)
        os.environ.setdefault("JAX_PLATFORMS", "cpu")
        import jax
        from differt.geometry import rays_intersect_triangles  # the public operator of configs[1]

        f = jax.jit(lambda a, b, c: rays_intersect_triangles(a[:, None, :], b[:, None, :], c))
        jax.block_until_ready(f(o, d, tv))
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 5.0:
            jax.block_until_ready(f(o, d, tv))
            reps += 1
        out["reference_tests_per_s"] = o.shape[0] * tv.shape[0] * reps / (time.perf_counter() - t0)
    except Exception as exc:  # noqa: BLE001
        out["reference_error"] = repr(exc)[:200]
    return out


def cpu_baseline(num_triangles: int, budget_s: float = 12.0):
    """Time the CPU oracle (dense MT, OpenMP over rays) on a bounded sample of the workload."""
    import oracle as orc

    kind_note = "oracle/differt_oracle.c (C restatement), gcc -O3 -march=native -fopenmp, -ffp-contract=off"
    try:
        L = orc.lib(orc.build(native=True))
    except Exception:  # noqa: BLE001 - no compiler on the box: use the prebuilt generic oracle
        L = orc.lib()
        kind_note = "oracle/differt_oracle.c prebuilt (generic x86-64), -fopenmp, -ffp-contract=off"
    cores = os.cpu_count() or 1
    rs = 8192
    o, d, tv = make_cfg2(rs, num_triangles, seed=4321)
    t = np.empty((rs, num_triangles), np.float32)
    h = np.empty((rs, num_triangles), np.uint8)
    eps = C.c_float(10.0 * float(np.finfo(np.float32).eps))

    def run():
        L.orc_ray_intersect_triangle_dense(
            orc._p(o), orc._p(d), rs, orc._p(tv), num_triangles, eps, orc._p(t), orc._p(h)
        )

    run()  # warm-up (page faults)
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and reps >= 3) or reps >= 200:
            break
    return {
        "value": rs * num_triangles * reps / el,
        "unit": "ray-triangle tests/s",
        "cores": cores,
        "kind": "port",
        **reference_probe(o[:256], d[:256], tv),
        "sample": f"{reps} passes of {rs} rays x {num_triangles} triangles (dense MT, same distribution), "
        f"{el:.1f} s; {kind_note}",
        "sample_short": f"{reps} x ({rs} rays x {num_triangles} triangles), {el:.1f} s",
    }


LINE_LIMIT = 4000  # characters of the one JSON line (the driver's stdout tail holds ~8 KB)


def _r(x, digits: int = 5):
    """Round floats to `digits` significant digits for the compact line (the sidecar keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _get(d, *keys):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(full: dict, sidecar: str | None) -> dict:
    """The one line the driver parses: the contract keys, `roofline`, `cpu_baseline`, `paths_metric` (the second
    half of BASELINE.json's metric) and a flat `paths` summary.  Numbers only; the legs in full are in `sidecar`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: _r(full[k], 7) for k in keep}
    out["config"] = {"workload": "configs[1]: dense ray_intersect_triangle fwd, rays x 10k random triangles",
                     "rays_per_gpu": full["config"]["rays_per_gpu"], "triangles": full["config"]["triangles"],
                     "rays_per_gpu_note": "ray axis extended from 256 (launch-bound) to fill one launch"}
    rf = full["roofline"]
    out["roofline"] = {k: _r(rf.get(k), 6) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "valu_frac",
                                                     "valu_frac_of_157TF", "traffic", "kernel_ms", "pmc_stale")}
    lit = full.get("cfg2_literal")
    if isinstance(lit, dict):  # the literal 256-ray launch of configs[1] (latency-bound), next to the filled launch
        out["cfg2_literal_us"] = _r(lit.get("us_per_launch_hipgraph") or lit.get("us_per_launch_back_to_back"))
        out["cfg2_literal_hbm_frac"] = _r(lit.get("hbm_frac"))
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb.get("sample_short", ""), "reference_installed": cb.get("reference_installed")}
        if cb.get("reference_tests_per_s"):
            out["cpu_baseline"]["reference_tests_per_s"] = _r(cb["reference_tests_per_s"])
    p = full.get("paths")
    if isinstance(p, dict) and "error" not in p:
        beam = p.get("beam_pruned") if isinstance(p.get("beam_pruned"), dict) else {}
        if beam.get("s_per_step"):
            # configs[2] through the full-coverage pruned search (the product's tracer for this size): fwd + grad(TX)
            out["paths_metric"] = {"value": _r(beam["valid_paths_per_s"]), "unit": "valid order-2 paths/s (fwd+grad)",
                                   "config": "configs[2]", "s_per_step": _r(beam["s_per_step"]),
                                   "valid_paths": beam["valid_paths"], "entry": "drt_trace_paths_beam",
                                   "same_as_exhaustive": beam.get("same_valid_paths_as_exhaustive")}
        else:
            out["paths_metric"] = {"value": _r(p.get("valid_paths_per_s")), "unit": "valid order-2 paths/s (fwd+grad)",
                                   "config": "configs[2]", "s_per_step": _r(p.get("s_per_step")),
                                   "valid_paths": p.get("valid_paths"), "entry": "drt_trace_paths_compact (rank range)"}
        same = [_get(p, leg, "same_valid_paths_as_exhaustive") for leg in ("beam_pruned", "beam_pruned_graph")]
        same3 = _get(p, "beam_pruned_order3", "same_valid_paths_as_exhaustive")
        same.append(same3.get("all_equal") if isinstance(same3, dict) else same3)
        summ = {
            "cfg2_exhaustive_s": p.get("s_per_step"), "cfg2_candidates_per_s": p.get("path_candidates_per_s"),
            "cfg2_filter_valu_frac": _get(p, "roofline", "frac"),
            "cfg2_filter_frac_of_157TF": _get(p, "roofline", "frac_of_157TF"),
            "cfg2_fwdgrad_s": beam.get("s_per_step"), "cfg2_valid_paths_per_s": beam.get("valid_paths_per_s"),
            "cfg2_graph_s": _get(p, "beam_pruned_graph", "s_per_step"),
            "cfg3_s": _get(p, "beam_pruned_order3", "s_per_step"),
            "cfg3_valid_paths": _get(p, "beam_pruned_order3", "valid_paths"),
            "cfg3_graph_s": _get(p, "beam_pruned_order3_graph", "s_per_step"),
            "cfg3_expand_ms": _get(p, "beam_pruned_order3", "kernel_ms", "last_expansion"),
            "cfg3_expand_valu_frac": _get(p, "beam_pruned_order3", "roofline", "frac"),
            "cfg3_expand_frac_of_157TF": _get(p, "beam_pruned_order3", "roofline", "frac_of_157TF"),
            "cfg3_pair_mode": _get(p, "beam_pruned_order3", "coplanar_pair_mode"),
            "dense_api_frac": _get(p, "reference_api", "roofline", "frac"),
            "dense_api_written_frac": _get(p, "reference_api", "roofline", "written_frac"),
            "dense_api_kernel_ms": _get(p, "reference_api", "roofline", "kernel_ms"),
            "same_as_exhaustive": all(bool(x) for x in same if x is not None) if any(x is not None for x in same) else None,
            "cpu_candidates_per_s": _get(p, "cpu_baseline", "value"),
        }
        real = p.get("real_meshes")
        if isinstance(real, dict) and "error" not in real:
            summ["bruxelles_order2_s"] = _get(real, "bruxelles", "beam_order2", "s_per_step")
            summ["bruxelles_order3_s"] = _get(real, "bruxelles", "beam_order3", "s_per_step")
            summ["bruxelles_order3_triangles_s"] = _get(real, "bruxelles", "beam_order3_triangles", "s_per_step")
            summ["bruxelles_pairs"] = _get(real, "bruxelles", "paired_primitives")
            summ["bruxelles_same_as_exhaustive"] = _get(real, "bruxelles", "beam_order2", "same_valid_paths_as_exhaustive")
            b3 = _get(real, "bruxelles", "beam_order3", "same_valid_paths_as_exhaustive")
            summ["bruxelles_order3_pairs_checked"] = b3.get("checked_pairs") if isinstance(b3, dict) and b3.get("all_equal") else None
            c3 = _get(p, "beam_pruned_order3", "same_valid_paths_as_exhaustive")
            summ["cfg3_pairs_checked"] = c3.get("checked_pairs") if isinstance(c3, dict) and c3.get("all_equal") else None
        out["paths"] = {k: _r(v) for k, v in summ.items()}
    elif isinstance(p, dict):
        out["paths"] = {"error": str(p["error"])[:200]}
    sc = full.get("strong_scaling")
    if isinstance(sc, dict) and "error" not in sc:
        out["paths"] = out.get("paths", {})
        out["paths"]["cfg4_s"] = _r(_get(sc, "beam_sharded", "s_per_step"))
        out["paths"]["cfg4_valid_paths"] = _get(sc, "beam_sharded", "valid_paths")
        out["paths"]["cfg4_graph_s"] = _r(_get(sc, "beam_graph", "s_per_step"))
        out["ranks_seen_by_backend"] = sc.get("ranks_seen_by_backend")
    if "strong_headline" in full:
        out["strong_headline"] = [{k: _r(v) for k, v in h.items()} for h in full["strong_headline"]]
    if full.get("legs_error"):
        out["legs_error"] = str(full["legs_error"])[:160]
    out["full"] = sidecar
    return out


def fit_line(comp: dict, limit: int = LINE_LIMIT) -> str:
    """The compact dict as ONE JSON line below `limit` characters.  A line that outgrew the limit sheds its optional blocks
    (they stay in the sidecar) -- never the contract keys, `roofline` or `cpu_baseline`, and never the run: an assert here
    once lost every measurement of a run whose line was a few characters too long."""
    line = json.dumps(comp, separators=(",", ":"))
    for drop in ("strong_headline", "paths", "legs_error", "paths_metric", "ranks_seen_by_backend"):
        if len(line) < limit:
            break
        if drop in comp:
            comp.pop(drop)
            comp["line_shed"] = comp.get("line_shed", []) + [drop]
            line = json.dumps(comp, separators=(",", ":"))
    return line


def _progress(msg: str) -> None:
    """Leg markers on stderr (the line on stdout stays the only thing there): a leg that dies is then named."""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def run_legs(args, dev, rank: int, world: int, dist, save=None) -> dict:
    """Every leg besides the headline: {"paths", "strong_scaling", "strong_headline" (N > 1), "queries"}.
    `save(dict)`: called with the legs finished so far (the child process of a 1-GPU run writes them to its JSON file after
    every leg, so that a leg that dies or outlives the time limit costs its own numbers only)."""
    out: dict = {}

    def keep():
        if save is not None and rank == 0:
            save(out)
    if not args.no_paths:  # every rank takes part: the candidate-rank space is sharded over the GPUs
        paths = None
        try:
            import bench_paths

            def partial(p):
                if save is not None and rank == 0:
                    save({**out, "paths": p})

            paths = bench_paths.run(dev, cpu_sample=not args.no_cpu_baseline, rank=rank, world=world,
                                    dist=dist, checkpoint=partial)
        except ImportError:
            pass
        except Exception as exc:  # noqa: BLE001 - the headline line must still be printed
            paths = {"error": repr(exc)}
        if rank == 0 and paths is not None:
            out["paths"] = paths

    _progress("paths done")
    keep()
    if not args.no_scaling:  # every rank takes part: configs[4], total work fixed, split over the ranks
        try:
            import bench_scaling

            sc = bench_scaling.run(dev, rank=rank, world=world, dist=dist, boxes=args.cfg5_boxes,
                                   window=args.cfg5_window, rx_side=args.cfg5_rx_side)
        except Exception as exc:  # noqa: BLE001
            sc = {"error": repr(exc)}
        if rank == 0:
            out["strong_scaling"] = sc
            if world > 1 and isinstance(sc, dict):
                # the north-star scaling claim (>= 6x at 8 GPUs) is about FIXED total work: surface those
                # legs at the top level so that a SCALE run shows them without digging (`value` above stays
                # the weak-scaling dense operator, identical to the single-GPU bench at N = 1)
                heads = []
                n1 = strong_n1_record()
                # triangle_block is the leg configs[4] literally names (T/N triangles per rank, ONE MIN all-reduce of packed
                # (t, tie) keys per batch); beam_sharded / candidate_sharded are the collective-free splits
                for leg in ("triangle_block", "beam_sharded", "candidate_sharded"):
                    rec = sc.get(leg)
                    if isinstance(rec, dict) and rec.get("s_per_step") is not None:
                        h = {"leg": leg, "s_per_step": rec["s_per_step"], "n_gpus": world, "scaling": "strong"}
                        if rec.get("valid_paths") is not None:
                            h["valid_paths"] = rec["valid_paths"]
                        if n1 and n1.get(leg):
                            h["speedup_vs_n1"] = n1[leg] / rec["s_per_step"]
                        heads.append(h)
                out["strong_headline"] = heads

    _progress("scaling done")
    keep()
    if rank == 0 and not args.no_paths:
        try:
            import bench_queries

            out["queries"] = bench_queries.run(dev)
        except Exception as exc:  # noqa: BLE001
            out["queries"] = {"error": repr(exc)}

    return out


def strong_n1_record() -> dict | None:
    """{leg: s_per_step} of the committed ONE-GPU run of the strong-scaling legs (profiles/r*/strong_n1.json, written by
    `bench.py --write-strong-n1`), or None when its source hashes do not describe the kernels in the tree."""
    recs = sorted((ROOT / "profiles").glob("r*/strong_n1.json"))
    if not recs:
        return None
    try:
        from differt_amd._srchash import source_hash

        rec = json.loads(recs[-1].read_text())
        if any(rec["source_hash"].get(k) != source_hash(k) for k in ("beam", "trace_filter", "dense")):
            return None
        return rec["s_per_step"]
    except Exception:  # noqa: BLE001
        return None


def legs_in_child(args) -> dict:
    """run_legs in a child process (one GPU): its JSON, or {"legs_error": ...} when the child died."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        outp = Path(tmp) / "legs.json"
        cmd = [sys.executable, str(Path(__file__).resolve()), "--legs-child", str(outp), "--cfg5-boxes", str(args.cfg5_boxes),
               "--cfg5-rx-side", str(args.cfg5_rx_side)]
        if args.cfg5_window is not None:
            cmd += ["--cfg5-window", str(args.cfg5_window)]
        for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-paths", args.no_paths), ("--no-scaling", args.no_scaling)):
            if on:
                cmd.append(flag)
        try:
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, timeout=1500)  # (its stderr = ours: the progress markers)
            if r.returncode == 0 and outp.exists():
                return json.loads(outp.read_text())
            err = (f"the legs process exited with code {r.returncode} (negative: killed by that signal); "
                   "the last [bench] marker on stderr names the leg")
        except subprocess.TimeoutExpired:
            err = "the legs process did not finish within 1500 s"
        done = {}
        try:  # what the child had finished (it rewrites its file after every leg)
            done = json.loads(outp.read_text()) if outp.exists() else {}
        except (OSError, ValueError):
            done = {}
        return {**done, "legs_error": err}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-json", default=None, help="sidecar with every leg in full (default gpurun_out/bench_full.json)")
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed launches after 20 warm-ups: the first tens of milliseconds after an idle period run at a
    # lower clock (a 20-step window measured 0.94 ms/step where steady state is 0.79 ms on the same box)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=65536, help="rays per GPU per step")
    ap.add_argument("--triangles", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paths", action="store_true")
    ap.add_argument("--no-scaling", action="store_true", help="skip the configs[4] strong-scaling legs")
    ap.add_argument("--cfg5-boxes", type=int, default=20000, help="boxes of the configs[4] city (10 triangles each)")
    ap.add_argument("--cfg5-window", type=int, default=None, help="candidate ranks per pair of the strong-scaling leg")
    ap.add_argument("--cfg5-rx-side", type=int, default=32)
    ap.add_argument("--legs-in-process", action="store_true", help="run the extra legs in this process even on one GPU")
    ap.add_argument("--write-strong-n1", default=None, metavar="JSON",
                    help="one GPU: write {leg: s_per_step} of the strong-scaling legs + source hashes (what speedup_vs_n1 of an "
                         "N > 1 run is computed against)")
    ap.add_argument("--legs-child", default=None, help=argparse.SUPPRESS)  # internal: run the legs only, write their JSON here
    args = ap.parse_args()

    import torch

    import differt_amd._lib as lib
    from differt_amd._tensors import ptr, stream

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    # Test hook: DRT_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo, so that the N > 1 code path
    # (sharding, barriers, the MAX / SUM all-reduces, rank-0 JSON) can be exercised on a 1-GPU box.
    # Numbers from such a run are meaningless and are tagged in the JSON.
    share_gpu = os.environ.get("DRT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    lib.require_device()
    dev = torch.device("cuda", local_rank)
    if args.legs_child:  # the child of legs_in_child(): legs only, JSON to the given path, nothing on stdout
        def save(d, path=Path(args.legs_child)):
            tmp = path.with_suffix(".tmp")
            tmp.write_text(json.dumps(d))
            tmp.replace(path)  # (atomic: the parent never reads half a file)

        save(run_legs(args, dev, 0, 1, None, save=save))
        return

    R, T = args.rays, args.triangles
    o_h, d_h, tv_h = make_cfg2(R, T, seed=1234 + rank)
    _, _, tv_h = make_cfg2(1, T, seed=1234)  # the triangle set is replicated on every rank
    o, d, tv = (torch.as_tensor(x, device=dev) for x in (o_h, d_h, tv_h))
    t_out = torch.empty((R, T), dtype=torch.float32, device=dev)
    hit_out = torch.empty((R, T), dtype=torch.uint8, device=dev)
    eps = 10.0 * float(np.finfo(np.float32).eps)

    def step():
        lib.call("drt_ray_intersect_triangle_dense", ptr(o), ptr(d), R, ptr(tv), T, eps, ptr(t_out),
                 ptr(hit_out), stream())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Time-based pre-warm BEFORE the counted warm-up: the metric is steady-state throughput, and the
    # first tens of milliseconds after an idle period run at a ramping clock (a cold 20-launch window
    # measured 0.94 ms/step where the same process settles at 0.79 ms).  Launch blocks of 50 until two
    # consecutive block means agree within 1 %, capped at 1 s of device time.
    prewarm_ms, prewarm_launches, prev_mean = 0.0, 0, None
    pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while prewarm_ms < 1000.0:
        pe0.record()
        for _ in range(50):
            step()
        pe1.record()
        pe1.synchronize()
        blk = pe0.elapsed_time(pe1)
        prewarm_ms += blk
        prewarm_launches += 50
        if prev_mean is not None and abs(blk - prev_mean) <= 0.01 * prev_mean:
            break
        prev_mean = blk
    for _ in range(args.warmup):
        step()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()  # recorded on torch's current stream == the stream handed to the C ABI
        step()
        ev1[i].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))

    # light self-check so that a broken kernel cannot post a number
    nhit = int(hit_out[:256].sum().item())
    finite = bool(torch.isfinite(t_out[:4]).any().item())
    assert finite and 0 <= nhit < 256 * T

    result = None
    if rank == 0:
        tests_per_step = R * T * world
        algo_bytes = 5 * R * T + 24 * R + 36 * T  # 5 B out/test + inputs once
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch come from a committed counter pass (rocprofv3 --pmc cannot run inside this
        # process); the record carries a hash of the kernel sources it was collected on, and a record that
        # no longer describes the tree is flagged instead of being quoted silently
        traffic, pmc_stale = None, None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                from differt_amd._srchash import is_stale

                rec = json.loads(pmc.read_text())
                traffic = rec.get("mt_dense_kernel_bytes_per_launch")
                pmc_stale = is_stale(rec, "dense")
            except Exception:  # noqa: BLE001
                traffic, pmc_stale = None, None
        result = {
            "metric": "ray-triangle tests/s (ray_intersect_triangle dense fwd, 10k random triangles)",
            "value": tests_per_step * args.steps / elapsed,
            "unit": "ray-triangle tests/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_ms": prewarm_ms,
            "prewarm_launches": prewarm_launches,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (TEST HOOK: all ranks share cuda:0 over gloo, numbers invalid)" if share_gpu else ""),
            "config": {
                "workload": "BASELINE configs[1] (cfg2): rays x 10k random triangles, dense "
                "ray_intersect_triangle fwd; ray axis extended from 256 to --rays (same seeded "
                "distribution) so that one launch moves GBs, rays sharded over GPUs",
                "rays_per_gpu": R,
                "triangles": T,
                "outputs": "t f32[R,T] + hit u8[R,T]",
            },
            "roofline": {
                "kernel": "drt::mt_dense_aligned_kernel",
                # co-limited: 5 B written per test against 8 TB/s AND 58.5 one-rounding VALU instructions
                # per test (bit-identical to a no-FMA oracle) against the non-FMA issue ceiling of 256 CUs x 4 SIMDs x 32
                # lanes x 2.4 GHz = 7.86e13 lane-operations/s (= half of the 157.3 TFLOP/s FP32 vector peak, which counts FMAs)
                "bound": "hbm+valu",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "valu_frac": VALU_PER_TEST * R * T / (kernel_ms * 1e-3) / VALU_ISSUE_PEAK,
                "valu_frac_of_157TF": VALU_PER_TEST * R * T / (kernel_ms * 1e-3) / 157.3e12,
                "valu_instructions_per_test": VALU_PER_TEST,
                "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2)",
                "pmc_stale": pmc_stale,
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }

    # literal configs[1]: 256 rays (launch-bound), rank 0 only, outside the timed region
    if rank == 0:
        Rl = 256
        ol, dl, _ = make_cfg2(Rl, T, seed=99)
        ol, dl = torch.as_tensor(ol, device=dev), torch.as_tensor(dl, device=dev)
        tl = torch.empty((Rl, T), dtype=torch.float32, device=dev)
        hl = torch.empty((Rl, T), dtype=torch.uint8, device=dev)

        def step_l():
            lib.call("drt_ray_intersect_triangle_dense", ptr(ol), ptr(dl), Rl, ptr(tv), T, eps,
                     ptr(tl), ptr(hl), stream())

        for _ in range(10):
            step_l()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            step_l()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        # the same 200 launches captured in one HIP graph (the entry point is capturable: no sync,
        # no allocation): removes the ~10 us/launch of Python + ctypes from the loop
        us_graph = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    step_l()
            g.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us_graph = e0.elapsed_time(e1) * 1e3 / (5 * n)
        except Exception:  # noqa: BLE001 - graph capture is an extra, never fatal for the bench line
            us_graph = None
        result["cfg2_literal"] = {
            "rays": Rl,
            "triangles": T,
            "us_per_launch_back_to_back": us,
            "us_per_launch_hipgraph": us_graph,
            "tests_per_s": Rl * T / ((us_graph or us) * 1e-6),
            "hbm_frac": (5 * Rl * T + 24 * Rl + 36 * T) / ((us_graph or us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
        }

    # configs[1] as DiffeRT issues it under leading batch axes: 64 problems of 256 rays x 10k triangles
    # (own triangle set each) in ONE launch of the batched entry point
    if rank == 0:
        try:
            Bn, Rl = 64, 256
            ob, db, _ = make_cfg2(Bn * Rl, T, seed=77)
            tvb = np.stack([make_cfg2(1, T, seed=500 + b)[2] for b in range(Bn)])
            ob, db, tvb = (torch.as_tensor(x, device=dev) for x in (ob, db, tvb))
            tb = torch.empty((Bn, Rl, T), dtype=torch.float32, device=dev)
            hb = torch.empty((Bn, Rl, T), dtype=torch.uint8, device=dev)

            def step_b():
                lib.call("drt_ray_intersect_triangle_dense_batched", ptr(ob), ptr(db), 3 * Rl, Rl, ptr(tvb), 9 * T, T,
                         Bn, eps, ptr(tb), ptr(hb), stream())

            for _ in range(20):
                step_b()
            nbt = 100
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(nbt):
                step_b()
            e1.record()
            torch.cuda.synchronize()
            msb = e0.elapsed_time(e1) / nbt
            bytes_b = Bn * (5 * Rl * T + 24 * Rl + 36 * T)
            result["cfg2_batched"] = {
                "problems": Bn, "rays": Rl, "triangles": T, "ms_per_launch": msb,
                "tests_per_s": Bn * Rl * T / (msb * 1e-3),
                "hbm_frac": bytes_b / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
            }
            del ob, db, tvb, tb, hb
        except Exception as exc:  # noqa: BLE001 - an extra leg never costs the headline line
            result["cfg2_batched"] = {"error": repr(exc)}

    _progress("headline done")
    # The other legs (image-method traces, real meshes, configs[4], queries).  On ONE GPU they run in a child process: a
    # leg that takes the process down (a GPU memory fault aborts it -- round 5 met one, in the bench's own graph leg)
    # then costs its own numbers, not the headline line.  Under torch.distributed (N > 1) they run here: they shard
    # over the ranks and use the process group.
    if world == 1 and not args.legs_in_process:
        del t_out, hit_out
        torch.cuda.empty_cache()
        result.update(legs_in_child(args))
    else:
        legs = run_legs(args, dev, rank, world, dist)  # (every rank takes part; rank 0 holds the record)
        if rank == 0:
            result.update(legs)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(T)

    barrier()
    if rank == 0:
        # The driver keeps ~8 KB of stdout: the LINE is a compact summary (< 4 000 characters, no prose);
        # every leg in full goes to a sidecar file whose path the line carries.
        side = Path(args.full_json) if args.full_json else ROOT / "gpurun_out" / "bench_full.json"
        try:
            side.parent.mkdir(parents=True, exist_ok=True)
            side.write_text(json.dumps(result, indent=1))
            side_name = str(side.relative_to(ROOT)) if side.is_relative_to(ROOT) else str(side)
        except OSError:
            side_name = None
        sc1 = result.get("strong_scaling")
        if args.write_strong_n1 and world == 1 and isinstance(sc1, dict):
            from differt_amd._srchash import source_hash

            n1 = {leg: sc1[leg]["s_per_step"] for leg in ("triangle_block", "beam_sharded", "candidate_sharded")
                  if isinstance(sc1.get(leg), dict) and sc1[leg].get("s_per_step") is not None}
            Path(args.write_strong_n1).write_text(json.dumps(
                {"what": "bench.py strong-scaling legs on ONE MI355X (fixed total work: configs[4])", "s_per_step": n1,
                 "source_hash": {k: source_hash(k) for k in ("beam", "trace_filter", "dense")}}, indent=1) + "\n")
        print(fit_line(compact_line(result, side_name)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
