"""Which buffer placement makes the dense tracer's kernel fast or slow?  (profiles/r04/dense.md: the same kernel ran at
0.79 or 1.04 ms per launch depending on which generation of the caching allocator's blocks held the outputs.)
One arena, the four outputs + the survivor queue carved at controlled relative offsets; kernel time from drt_trace_stats.
python scratch/dense_placement.py"""
import ctypes as C
import json
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
ntx, nrx, Cn, k = 1, 64, 1 << 20, 2
table = torch.empty((Cn, k), dtype=torch.int32, device="cuda")
_lib.call("drt_candidates_fill", 10000, k, 0, Cn, None, 1, ptr(table), stream())
rows = ntx * nrx * Cn
sizes = {"v": rows * 48, "o": rows * 16, "t": rows * 8, "m": rows, "ws": 64 + rows * 8}
arena = torch.empty(16 << 30, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
base += (-base) % (2 << 20)
lib = _lib.load()
h = mesh.handle().h
cands = _table_candidates(table)


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
    ts = sorted(ts[1:])
    return ts[len(ts) // 2]


def layout(pad):
    """arrays back to back, each start rounded up to 2 MiB, plus pad[n] extra bytes before array n"""
    off, out = 0, {}
    for n in ("v", "o", "t", "m", "ws"):
        off += (-off) % (2 << 20)
        off += pad.get(n, 0)
        out[n] = off
        off += sizes[n]
    return out


res = []
del arena
arena2 = torch.empty(100 << 30, dtype=torch.uint8, device="cuda")
base = arena2.data_ptr()
base += (-base) % (1 << 30)
G16 = 16 << 30
res.append(("packed", run(layout({}))))
res.append(("objects.. +16 GiB", run(layout({"o": G16}))))
res.append(("objects.. +32 GiB", run(layout({"o": 2 * G16}))))
res.append(("types.. +16 GiB", run(layout({"t": G16}))))
res.append(("mask.. +16 GiB", run(layout({"m": G16}))))
res.append(("every array +16 GiB after the previous", run(layout({"o": G16, "t": G16, "m": G16, "ws": G16}))))
res.append(("objects.. +12 GiB", run(layout({"o": 12 << 30}))))
res.append(("objects.. +15 GiB", run(layout({"o": 15 << 30}))))
res.append(("objects.. +17 GiB", run(layout({"o": 17 << 30}))))
print(json.dumps(res, indent=1))
