"""Lab: row-block order of the dense tracer's kernel (default: one contiguous row range per XCD; DRT_DENSE_LAB_MODE=1:
round 4's interleaved order) against the placement of the outputs (profiles/r04/dense.md: packed = slow level,
objects.. +32 GiB = fast level).  The first version of this script also tried a receiver loop skewed per block and row
blocks strided by 1031: the same few per cent as the XCD order, no better combined (table in dense.md).
python scratch/dense_modes.py [--chunk=N]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from differt_amd import _lib  # noqa: E402
from differt_amd._tensors import ptr, stream  # noqa: E402
from differt_amd.geometry._solvers import _params, _table_candidates  # noqa: E402

V, Tr, c, h = S.manhattan(1000)
tx, rx = S.manhattan_tx_rx(c, h, 16, 64)
mesh = G.Mesh(V, Tr)
txd, rxd = torch.tensor(tx[:1], device="cuda"), torch.tensor(rx, device="cuda")
CN = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--chunk=")), str(1 << 20)))
ntx, nrx, k = 1, 64, 2
table = torch.empty((CN, k), dtype=torch.int32, device="cuda")
_lib.call("drt_candidates_fill", 10000, k, 0, CN, None, 1, ptr(table), stream())
rows = ntx * nrx * CN
sizes = {"v": rows * 48, "o": rows * 16, "t": rows * 8, "m": rows, "ws": 64 + rows * 8}
arena = torch.empty(48 << 30, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
base += (-base) % (1 << 30)
h = mesh.handle().h
cands = _table_candidates(table)
ref = {}


def run(offsets, reps=6):
    st = _lib.TraceStats()
    params = _params(None, None, None)
    params.stats = C.pointer(st)
    p = {n: base + offsets[n] for n in sizes}
    ts = []
    for _ in range(reps):
        _lib.call("drt_trace_paths_dense_ex", h, C.byref(params), ptr(txd), ntx, ptr(rxd), nrx, C.byref(cands), None,
                  C.c_void_p(p["v"]), C.c_void_p(p["o"]), C.c_void_p(p["m"]), C.c_void_p(p["t"]), C.c_void_p(p["ws"]),
                  sizes["ws"], stream())
        ts.append(st.filter_ms)
    torch.cuda.synchronize()
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], int(st.valid)


def layout(pad):
    off, out = 0, {}
    for n in ("v", "o", "t", "m", "ws"):
        off += (-off) % (2 << 20)
        off += pad.get(n, 0)
        out[n] = off
        off += sizes[n]
    return out


res = []
for mode in (1, 0, 1, 0):
    os.environ["DRT_DENSE_LAB_MODE"] = str(mode)
    packed, nv = run(layout({}))
    far, nv2 = run(layout({"o": 32 << 30}))
    res.append({"mode": mode, "packed_ms": packed, "objects_plus_32GiB_ms": far, "valid": nv, "valid_far": nv2})
    print(json.dumps(res[-1]), flush=True)
