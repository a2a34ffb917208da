"""Order-3 pruned search on the reference's meshes + configs[3]: step time, levels, rows (A/B of margin constants)."""
import json, sys
sys.path.insert(0, ".")
import bench_paths as BP
import differt_amd.geometry as G
import synthetic_scenes as S
for name in ("manhattan", "manhattan_small", "bruxelles"):
    V, Tr = S.load_real_mesh(name)
    tx, rx = S.outdoor_end_points(G, V, Tr, 16, 64)
    r = BP.beam_leg(G, G.Mesh(V, Tr), tx, rx, 3, None, reps=2)
    print(json.dumps({name: {k: r.get(k) for k in ("s_per_step", "valid_paths", "rows_traced", "prefix_levels", "kernel_ms", "error")}}), flush=True)
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
r = BP.beam_leg(G, G.Mesh(V, Tr), tx, rx, 3, None, reps=2)
print(json.dumps({"cfg3": {k: r.get(k) for k in ("s_per_step", "valid_paths", "rows_traced", "prefix_levels", "kernel_ms", "error")}}), flush=True)
