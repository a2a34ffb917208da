"""The library's own capture-safe radix sort (csrc/radix_sort.hip, drt_sort_u64): stable, bit ranges, payloads, sizes
around the tile / chunk boundaries, and three replays inside a HIP graph (the reason it exists: library radix sorts reset
their state with memset nodes, which do not replay reliably on this ROCm -- tests/test_hipgraph_gpu.py)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(keys, vals, begin, end):
    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream

    n = keys.numel()
    wb = _lib.load().drt_sort_u64_workspace_size(n, 0 if vals is None else 1)
    ws = torch.empty(max(wb, 1), dtype=torch.uint8, device="cuda")
    ko = torch.empty_like(keys)
    vo = None if vals is None else torch.empty_like(vals)
    _lib.call("drt_sort_u64", ptr(keys), ptr(ko), ptr(vals), ptr(vo), n, begin, end, ptr(ws), ws.numel(), stream())
    return ko, vo, ws


def _reference(keys: np.ndarray, begin: int, end: int):
    width = end - begin
    part = (keys >> np.uint64(begin)) & np.uint64((1 << width) - 1 if width < 64 else 0xFFFFFFFFFFFFFFFF) if width else np.zeros_like(keys)
    return np.argsort(part, kind="stable")


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2047, 2048, 2049, 5000, 1 << 15, (1 << 15) + 1, 300_001, 1 << 21])
@pytest.mark.parametrize("bits", [(0, 64), (0, 63), (0, 42), (3, 29), (0, 8), (5, 5), (40, 64)])
def test_sort_matches_a_stable_reference(n, bits):
    rng = np.random.default_rng(n * 131 + bits[1])
    # few distinct values in the sorted bit range as well: stability is then visible through the payload
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    if n % 2:
        keys &= np.uint64(0x00FF00FF00FF00FF)
    keys[rng.random(n) < 0.05] = np.uint64(0xFFFFFFFFFFFFFFFF)  # padding keys (-1 as int64)
    vals = np.arange(n, dtype=np.uint32)
    tk = torch.from_numpy(keys.view(np.int64)).cuda()
    tv = torch.from_numpy(vals.view(np.int32)).cuda()
    ko, vo, _ = _sort(tk, tv, *bits)
    perm = _reference(keys, *bits)
    assert np.array_equal(vo.cpu().numpy().view(np.uint32), vals[perm])
    assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys[perm])
    assert np.array_equal(tk.cpu().numpy().view(np.uint64), keys)  # the input is left intact
    ko2, _, _ = _sort(tk, None, *bits)  # keys only
    assert torch.equal(ko2, ko)


def test_sort_replays_in_a_hip_graph():
    """Three replays with different inputs: every replay is a correct sort (memset-free by construction)."""
    from differt_amd import _lib
    from differt_amd._tensors import ptr, stream

    n = 1_500_000
    rng = np.random.default_rng(5)
    tk = torch.zeros(n, dtype=torch.int64, device="cuda")
    tv = torch.arange(n, dtype=torch.int32, device="cuda")
    ko, vo = torch.empty_like(tk), torch.empty_like(tv)
    wb = _lib.load().drt_sort_u64_workspace_size(n, 1)
    ws = torch.empty(wb, dtype=torch.uint8, device="cuda")

    def launch():
        _lib.call("drt_sort_u64", ptr(tk), ptr(ko), ptr(tv), ptr(vo), n, 0, 50, ptr(ws), ws.numel(), stream())

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for rep in range(3):
        keys = rng.integers(0, 1 << 50, n, dtype=np.uint64)
        tk.copy_(torch.from_numpy(keys.view(np.int64)))
        g.replay()
        torch.cuda.synchronize()
        perm = np.argsort(keys, kind="stable")
        assert np.array_equal(ko.cpu().numpy().view(np.uint64), keys[perm]), rep
        assert np.array_equal(vo.cpu().numpy().view(np.uint32), perm.astype(np.uint32)), rep


def test_sort_throughput_is_reported():
    """2^25 keys, 63 bits (the row sort of the asynchronous pruned tracer at configs[4] scale): time per sort, for the record."""
    n = 1 << 25
    tk = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    _sort(tk, None, 0, 63)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ko, _, _ = _sort(tk, None, 0, 63)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    assert bool((ko[1:] >= ko[:-1]).all())
    print(f"drt_sort_u64: 2^25 keys x 63 bits in {ms:.2f} ms ({n / ms / 1e6:.1f} G keys/s)")
    assert ms < 50.0
