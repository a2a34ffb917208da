"""Quick timing of the pruned search on the BASELINE scenes (the legs of bench_paths.py / bench_real.py / bench_scaling.py that
matter when a margin constant or an expansion kernel changes): configs[2], configs[3], bruxelles order 2 / 3, configs[4].
python scratch/time_beam_r06.py [--skip4]"""
import json
import sys

sys.path.insert(0, ".")
import bench_paths as BP  # noqa: E402
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402

out = {}
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
mesh = G.Mesh(V, Tr)
for order, reps in ((2, 5), (3, 3)):
    r = BP.beam_leg(G, mesh, tx, rx, order, None, reps=reps)
    out[f"cfg{order}"] = {k: r.get(k) for k in ("s_per_step", "valid_paths", "rows_traced", "prefix_levels", "kernel_ms", "error")}
    print(json.dumps({f"cfg{order}": out[f"cfg{order}"]}), flush=True)
V, Tr = S.load_real_mesh("bruxelles")
tx, rx = S.outdoor_end_points(G, V, Tr, 16, 64)
mesh = G.Mesh(V, Tr)
for order, reps in ((2, 5), (3, 3)):
    r = BP.beam_leg(G, mesh, tx, rx, order, None, reps=reps)
    out[f"bruxelles{order}"] = {k: r.get(k) for k in ("s_per_step", "valid_paths", "rows_traced", "prefix_levels", "kernel_ms", "error")}
    print(json.dumps({f"bruxelles{order}": out[f"bruxelles{order}"]}), flush=True)
if "--skip4" not in sys.argv:
    V, Tr, tx, rx = S.cfg5_scene()
    mesh = G.Mesh(V, Tr)
    r = BP.beam_leg(G, mesh, tx, rx, 2, None, reps=3)
    out["cfg4"] = {k: r.get(k) for k in ("s_per_step", "valid_paths", "rows_traced", "prefix_levels", "kernel_ms", "error")}
    print(json.dumps({"cfg4": out["cfg4"]}), flush=True)
