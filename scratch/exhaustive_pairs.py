"""Completeness of the pruned ("beam") search at the BASELINE configs' OWN size, against the exhaustive tracer.

    python scratch/exhaustive_pairs.py [--pairs3 4] [--pairs5 4] [--pairs-real 8] [--out profiles/r06/stress/exhaustive_pairs.json]

configs[3] (order 3, 10 000 triangles): for each selected (tx, rx) pair, `trace_rank_range` evaluates ALL
n(n-1)^2 = 9.998e11 candidates of that pair (reference: full enumeration, geometry/_solvers.py:803-848 +
_trace_path_candidates :499-770) in rank windows; the objects / vertex bits it finds must equal the rows of the
full 16 x 64 beam result (kappa = 64 AND kappa = 1) that belong to the pair.  configs[4] (order 2, 200 000
triangles, 1 x 1024): the same with n(n-1) = 4.0e10 candidates per pair.

Pair selection (deterministic): the pairs with the most paths in the beam result, pairs with none, and the
"grazing-heavy" pairs -- end points with the most triangle planes passing within 1/180 of their distance (the
incidences that widen the search's margins, profiles/r03/beam.md).
"""


from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))

import synthetic_scenes as S  # noqa: E402
from exhaustive_pairs_check import check_config  # noqa: E402  (the checker itself lives with the tests that use it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs3", type=int, default=4)
    ap.add_argument("--pairs5", type=int, default=4)
    ap.add_argument("--pairs-real", type=int, default=0,
                    help="whole order-3 pair spaces on the reference's own meshes (bruxelles: 2.87e12 candidates each, manhattan)")
    ap.add_argument("--real-meshes", default="bruxelles,manhattan")
    ap.add_argument("--out", default="profiles/r06/stress/exhaustive_pairs.json")
    a = ap.parse_args()
    import differt_amd.geometry as G
    from differt_amd._srchash import source_hash
    out = {"what": "beam-pruned search == exhaustive tracer on whole candidate spaces of single (tx, rx) pairs at the "
                   "BASELINE configs' own size: objects and vertex bits", "records": [],
           # the kernels this record was taken on: bench_paths.exhaustive_record() quotes it only while these match the tree
           "source_hash": {k: source_hash(k) for k in ("beam", "trace_filter")}}
    if a.pairs3:
        V, Tr, c, h = S.manhattan(1000)
        tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
        out["records"].append(check_config("configs[3]", V, Tr, tx, rx, 3, a.pairs3, (64.0, 1.0), 64, 1 << 24))
    if a.pairs5:
        V, Tr, tx, rx = S.cfg5_scene()
        out["records"].append(check_config("configs[4]", V, Tr, tx, rx, 2, a.pairs5, (64.0, 1.0), 4, 1 << 24))
    if a.pairs_real:
        # the reference's benchmark mesh (differt/tests/benchmarks/fixtures.py:43-68) and manhattan.obj, the end points of
        # bench_real.py / tests/test_real_meshes_gpu.py: 16 TX x 64 RX in the open
        for name in a.real_meshes.split(","):
            V, Tr = S.load_real_mesh(name)
            tx, rx = S.outdoor_end_points(G, V, Tr, 16, 64)
            out["records"].append(check_config(f"{name} order 3", V, Tr, tx, rx, 3, a.pairs_real, (64.0, 1.0), 128, 1 << 24))
    out["all_equal"] = all(r["all_equal"] for r in out["records"])
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({"all_equal": out["all_equal"], "out": a.out}))
    if not out["all_equal"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
