"""Bounded slices of the randomized GPU-vs-oracle stress drivers (scratch/*_stress.py) inside the
`-m gpu` suite, plus run-to-run determinism of the cfg3 trace (hard part 6).

Each driver draws random scenes for a fixed wall-clock budget and counts mismatches; the bar is the
same as everywhere else: hit flags / indices / masks / objects bit-exact, `t` and path vertices bit
patterns identical (the full-length runs are recorded under profiles/)."""

from __future__ import annotations

import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
BUDGET_S = 12  # per driver; the whole module stays under ~90 s


def _run(script: str) -> dict:
    r = subprocess.run([sys.executable, f"scratch/{script}", str(BUDGET_S)], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_dense_moller_trumbore_vs_oracle_stress():
    st = _run("oracle_stress.py")
    assert st["cases"] > 0 and st["tests"] > 1e6 and st["hits"] > 0
    assert st["hit_mismatch"] == 0 and st["t_mismatch"] == 0, st


def test_any_hit_first_hit_vs_oracle_stress():
    st = _run("query_oracle_stress.py")
    assert st["cases"] > 0 and st["tests"] > 1e6 and st["hits"] > 0
    assert st["any_mismatch"] == 0 and st["idx_mismatch"] == 0 and st["t_mismatch"] == 0, st


def test_fused_tracer_vs_oracle_stress():
    st = _run("trace_oracle_stress.py")
    assert st["cases"] > 5 and st["candidate_evals"] > 1e4
    assert st["mask_mismatch"] == 0 and st["vertex_mismatch"] == 0 and st["object_mismatch"] == 0, st
    assert st["compact_mismatch"] == 0, st
    assert st.get("rotated", 0) > 1  # half of the box cities are rotated (round 5): not only axis-aligned geometry
    assert st.get("soups", 0) > 2 and st.get("soup_rows_near_valid_paths", 0) > 0  # round 6: triangle soups, rows around real paths


def test_hybrid_candidate_space_stress():
    st = _run("hybrid_stress.py")
    assert st["cases"] > 5
    assert st["object_mismatch_cases"] == 0 and st["vertex_mismatch_cases"] == 0 and st["pairs_not_subset"] == 0, st


def test_beam_pruning_vs_exhaustive_stress():
    """the conservatively pruned search returns exactly the exhaustive tracer's paths, and
    all expansion mappings / receiver stages agree on the candidate rows."""
    st = _run("beam_stress.py")
    assert st["cases"] > 50 and st["valid_paths"] > 0 and st["mapping_checks"] > 0
    assert st["rows_traced"] < st["exhaustive_candidates"] / 10
    assert st["missed"] == 0 and st["extra"] == 0 and st["vertex_mismatch"] == 0 and st["mapping_row_mismatch"] == 0, st
    assert st.get("rotated", 0) > 5 and st.get("shuffled", 0) > 5  # rotated cities, shuffled / thinned triangle arrays
    # round 6: half of the scenes are triangle soups -- gable / hip / ear-clipped roofs, slivers, T-junctions, 3-D rotations
    assert st.get("soups", 0) > 20 and st.get("soup_gable", 0) > 0 and st.get("soup_sliver_walls", 0) > 0 and st.get("soup_t_junctions", 0) > 0


def test_bvh_vs_brute_force_stress():
    r = subprocess.run([sys.executable, "scratch/bvh_stress.py", "20"], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    st = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    bad = {k: v for k, v in st.items() if "mismatch" in k and v}
    assert not bad, st


def test_cfg3_window_is_deterministic_run_to_run():
    """Two runs of the cfg3 step (16 TX x 64 RX, 10k-triangle city, order 2, fwd + grad) over a 5e6-rank
    window: keys, objects and vertex BITS identical (the radix sort removes the atomic append order);
    the gradients come from float atomics, so only their difference is bounded and reported."""
    import differt_amd.geometry as G
    import synthetic_scenes as S

    V, Tr, centres, heights = S.manhattan(1000)
    tx, rx = S.manhattan_tx_rx(centres, heights, 16, 64)
    mesh = G.Mesh(V, Tr)

    def step():
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        vl = mesh.vertices.detach().clone().requires_grad_(True)  # the leaf (Mesh keeps a reshaped view)
        mv = mesh.with_vertices(vl)
        scene = G.Scene(txg, torch.tensor(rx, device="cuda"), mv)
        p = G.ExhaustivePathTracer().trace_rank_range_literal(scene, 2, 0, 5_000_000)
        torch.sqrt((torch.diff(p.vertices, dim=-2) ** 2).sum(-1)).sum().backward()
        return p, txg.grad.clone(), vl.grad.clone()

    a, ga, gva = step()
    b, gb, gvb = step()
    assert a.objects.shape[0] > 0
    assert torch.equal(a.keys, b.keys) and torch.equal(a.objects, b.objects)
    assert torch.equal(a.vertices.detach().view(torch.int32), b.vertices.detach().view(torch.int32))
    for x, y in ((ga, gb), (gva, gvb)):
        scale = float(x.abs().max()) + 1e-30
        diff = float((x - y).abs().max())
        assert diff <= 1e-5 * scale, (diff, scale)  # atomics may re-associate the few per-vertex sums


@pytest.mark.parametrize("route", ["rank_range", "beam", "dense"])
def test_deterministic_grad_is_bit_identical_run_to_run(route):
    """DRT_TRACE_DETERMINISTIC_GRAD (SURVEY.md section 7 hard part 6): gradients w.r.t. transmitters, receivers
    and mesh vertices are summed in a fixed order (stable sort by destination + ordered sums): three runs give
    the SAME BITS, and the result is within 1e-6 (of the largest entry) of the float-atomic version."""
    import differt_amd.geometry as G
    import synthetic_scenes as S

    V, Tr, centres, heights = S.manhattan(60 if route == "dense" else 400, seed=21)
    tx, rx = S.manhattan_tx_rx(centres, heights, 4, 24, seed=22)
    mesh = G.Mesh(V, Tr)

    def step(det):
        txg = torch.tensor(tx, device="cuda", requires_grad=True)
        rxg = torch.tensor(rx, device="cuda", requires_grad=True)
        vl = mesh.vertices.detach().clone().requires_grad_(True)
        scene = G.Scene(txg, rxg, mesh.with_vertices(vl))
        tracer = G.ExhaustivePathTracer(deterministic_grad=det)
        if route == "rank_range":
            p = tracer.trace_rank_range_literal(scene, 2, max_survivors=1 << 22)
        elif route == "beam":
            p = tracer.trace_beam_pruned(scene, 2)
        else:
            cands, types = tracer.generate_path_candidates(scene, 1)
            p = tracer.trace_path_candidates(scene, cands, types)
        # many paths share a transmitter / receiver / wall vertex: real accumulation
        # (dense layout: cotangents only on the valid paths -- rejected candidates include ill-conditioned chains whose
        # 1e7-sized terms cancel, which says nothing about the summation order being tested)
        w = torch.linspace(0.5, 1.5, p.vertices.numel(), device="cuda").reshape(p.vertices.shape)
        (p.vertices * w * p.mask.reshape(*p.mask.shape, 1, 1).float()).sum().backward()
        return int(p.mask.sum()), txg.grad.clone(), rxg.grad.clone(), vl.grad.clone()

    runs = [step(True) for _ in range(3)]
    assert runs[0][0] > 20
    for other in runs[1:]:
        for x, y in zip(runs[0][1:], other[1:]):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    atomic = step(False)
    for x, y in zip(runs[0][1:], atomic[1:]):
        scale = float(y.abs().max()) + 1e-30
        assert float((x - y).abs().max()) <= 1e-6 * scale, (route, float((x - y).abs().max()), scale)
