"""Debug-build counters of the clustered expansion (build with DRT_EXTRA_FLAGS="-DDRT_LAB -DBEAM_LAB_COUNT"): box tests, surviving
(prefix, cluster) pairs, pairs with at least one child, children -- per level, configs[2] / configs[3]."""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from differt_amd import _lib  # noqa: E402

L = _lib.load()
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
tr = G.ExhaustivePathTracer(accel="bvh")
buf = (C.c_ulonglong * 16)()
for order in (2, 3):
    L.drt_debug_beam_counts(buf, 1)
    tr.trace_beam_pruned(scene, order)
    L.drt_debug_beam_counts(buf, 1)
    b = list(buf)
    print(json.dumps({"order": order, "box_tests": b[0], "pairs": b[1], "pairs_with_children": b[2], "children": b[3],
                      "pair_frac": b[1] / max(b[0], 1), "fruitful_frac": b[2] / max(b[1], 1),
                      "children_per_pair": b[3] / max(b[1], 1), "pairs_eps_inf": b[4], "pairs_eps_gt_1m": b[5], "pairs_eps_0p1_1m": b[6], "subboxes_passing": b[7],
                      "subboxes_per_pair": b[7] / max(b[1], 1),
                      "passes_reaching_pyramid": [b[8], b[10], b[12]], "lanes_alive_there": [b[9], b[11], b[13]],
                      **tr.last_beam_stats}))
