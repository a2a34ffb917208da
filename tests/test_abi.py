"""The C-ABI library loads without a GPU and exports every symbol include/differt_amd.h declares;
host-only entry points work; device entry points fail LOUDLY (status code, no abort, no CPU
fallback) when there is no GPU.  CPU only."""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
import torch

from differt_amd import _lib


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    unbound = [s for s in declared if s not in _lib._SIGNATURES]
    assert not unbound, f"declared in the header but no ctypes signature: {unbound}"
    stale = [s for s in _lib._SIGNATURES if s not in declared]
    assert not stale, f"ctypes signature without a declaration in the header: {stale}"
    assert L.drt_abi_version() == _lib.ABI_VERSION == 7
    assert f"#define DRT_ABI_VERSION {_lib.ABI_VERSION}" in _lib.HEADER_PATH.read_text()


def test_struct_layouts_match_header():
    # drt_trace_params: 3 floats + int32 + stats pointer; drt_trace_stats: 3 x i64, 3 x f32, i32; drt_candidates: ptr, 3 x i64, ptr, 2 x i32, (ptr, i64) x 2, 3 x ptr;
    # drt_em_params: f64, i32, 3 x f32, i32, 3 x f32
    assert C.sizeof(_lib.TraceParams) == 24 and _lib.TraceParams.stats.offset == 16
    assert C.sizeof(_lib.TraceStats) == 40 and _lib.TraceStats.filter_ms.offset == 24
    assert C.sizeof(_lib.Candidates) == 104 and _lib.Candidates.pair_offsets.offset == 80
    assert _lib.Candidates.order.offset == 40 and _lib.Candidates.first_map.offset == 48
    assert _lib.Candidates.num_last.offset == 72
    assert C.sizeof(_lib.EmParams) == 40 and _lib.EmParams.rx_polarization.offset == 24
    # drt_beam_params: f32, i32, 7 x i64, ptr; drt_beam_stats: 4 + 4 x i64, 2 x f32, 2 x i32, 4 x f32 (tests/abi/abi_beam_example.cpp asserts the same)
    assert C.sizeof(_lib.BeamParams) == 72 and _lib.BeamParams.stats.offset == 64 and _lib.BeamParams.shard_rank.offset == 48
    assert C.sizeof(_lib.BeamStats) == 96 and _lib.BeamStats.unit_m.offset == 64 and _lib.BeamStats.pair_mode.offset == 72


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_fails_loudly_not_silently():
    L = _lib.load()
    assert L.drt_device_check() == _lib.DRT_E_NO_DEVICE
    assert b"no HIP device" in L.drt_last_error()
    with pytest.raises(_lib.DrtError):
        _lib.require_device()
    import differt_amd.geometry as G

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G.ray_intersect_triangle(np.zeros(3), np.ones(3), np.zeros((3, 3)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G.Mesh.box()


def test_argument_validation_returns_status_codes():
    L = _lib.load()
    # invalid arguments are rejected before any device work
    rc = L.drt_ray_intersect_triangle_dense(None, None, -1, None, 5, 0.0, None, None, None)
    assert rc == _lib.DRT_E_INVALID and b"negative" in L.drt_last_error()
    with pytest.raises(ValueError):
        _lib.call("drt_complete_graph_fill_host", 3, 3, 4, 4, 0, 5, 2, None)
    c, o = C.c_uint64(), C.c_int32()
    _lib.call("drt_complete_graph_count", 10000, 10000, 10001, 4, C.byref(c), C.byref(o))
    assert c.value == 10000 * 9999 and o.value == 0
    assert L.drt_first_triangle_hit_by_ray_workspace_size(1000) == 8000
    assert L.drt_trace_dense_workspace_size(2, 3, 10) == 64 + 2 * 3 * 10 * 8


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under differt_amd/ may reference it."""
    import pathlib

    root = pathlib.Path(_lib.__file__).resolve().parent
    offenders = []
    for p in list(root.rglob("*.py")) + list(root.rglob("*.hip")) + list(root.rglob("*.hpp")) + list(root.rglob("*.cpp")):
        text = p.read_text()
        if "import oracle" in text or "from oracle" in text or "differt_oracle" in text:
            offenders.append(str(p))
    assert not offenders, offenders


def test_workspace_size_queries_work_without_a_gpu():
    """Size queries are host arithmetic: callable on a CPU box, monotone in the capacities, never zero."""
    L = _lib.load()
    bp = _lib.BeamParams()
    a = L.drt_trace_beam_workspace_size(16, 64, 10000, 2, C.byref(bp), 1 << 16)
    b = L.drt_trace_beam_workspace_size(16, 64, 10000, 3, C.byref(bp), 1 << 16)
    assert 0 < a < b  # order 3 adds the level-2 prefix list
    bp.max_rows = 1 << 20
    bp.max_records = 1 << 20
    bp.max_entries = 1 << 20
    c = L.drt_trace_beam_workspace_size(16, 64, 10000, 3, C.byref(bp), 1 << 16)
    assert 0 < c < b
    assert L.drt_trace_beam_workspace_size(1, 1, 12, 0, None, 4) == L.drt_trace_compact_workspace_size(1, 4)
    # default capacities follow the scene: BASELINE configs[0]'s 12-triangle box at order 3 stays
    # under 64 MB (round 3: 5.5 GiB whatever the scene); the 10k-triangle configs keep the round-3 capacities
    box = L.drt_trace_beam_workspace_size(1, 1, 12, 3, None, 1 << 16)
    assert 0 < box < 64 << 20
    assert L.drt_trace_beam_workspace_size(1, 1, 12, 3, None, 64) < 1 << 20
    full = _lib.BeamParams()
    full.max_entries, full.max_records, full.max_rows, full.max_survivors = 1 << 26, 1 << 27, 1 << 26, 1 << 22
    for (ntx, nrx, n, order) in ((16, 64, 10000, 2), (16, 64, 10000, 3), (1, 1024, 200000, 2)):
        assert (L.drt_trace_beam_workspace_size(ntx, nrx, n, order, None, 1 << 16)
                == L.drt_trace_beam_workspace_size(ntx, nrx, n, order, C.byref(full), 1 << 16))
    mid = L.drt_trace_beam_workspace_size(4, 16, 1000, 3, None, 1 << 16)
    assert box < mid < b
    assert L.drt_trace_vjp_workspace_size(0, 2) > 0
    assert L.drt_trace_vjp_workspace_size(1000, 3) > L.drt_trace_vjp_workspace_size(1000, 1)
