"""Debug build only (DRT_EXTRA_FLAGS=-DDRT_FILTER_DEBUG, DIFFERT_AMD_LIB=.../libdiffert_amd_dbg.so): how often a
wave of trace_filter_kernel leaves the fast paths on a window of configs[2]."""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
import differt_amd._lib as lib
import differt_amd.geometry as G
import synthetic_scenes as S
V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
scene = G.Scene(torch.tensor(tx, device="cuda"), torch.tensor(rx, device="cuda"), G.Mesh(V, Tr))
tr = G.ExhaustivePathTracer()
L = C.CDLL(str(lib.LIB_PATH))
buf = (C.c_ulonglong * 8)()
tr.trace_rank_range_literal(scene, 2, 0, 2_000_000)
L.drt_debug_counts(buf, 1)
tr.trace_rank_range_literal(scene, 2, 0, 20_000_000)
L.drt_debug_counts(buf, 0)
it = buf[0]
print({"wave_iterations": it, "step_j0_guarded": buf[1] / it, "step_j1_guarded": buf[2] / it, "mt_literal": buf[3] / it,
       "rare_branch": buf[4] / it})
