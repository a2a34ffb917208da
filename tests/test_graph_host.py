"""Host-side candidate enumeration (C++ unranking / DFS in libdiffert_amd.so) vs the oracle's literal
restatement of the Rust odometer, and vs the reference's ordered tables.  CPU only.

Mirrors differt-core/tests/geometry/test_graph.py and the rstest tables of graph.rs:1207-1706.
"""

from __future__ import annotations

import warnings

import numpy as np
import pytest

import oracle as orc
from differt_amd.geometry._graph import CompleteGraph, DiGraph
from differt_amd.geometry import (
    generate_all_path_candidates,
    generate_all_path_candidates_chunks_iter,
    generate_all_path_candidates_iter,
)


def test_reference_tables(goldens):
    """differt/tests/geometry/test_utils.py:448-476; graph.rs:1488-1513 (ordered)."""
    for case in goldens["generate_all_path_candidates"]["cases"]:
        got = generate_all_path_candidates(case["num_primitives"], case["order"])
        assert list(got.shape) == case["shape"]
        np.testing.assert_array_equal(got, np.asarray(case["rows"], dtype=np.int32).reshape(case["shape"]))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 8])
@pytest.mark.parametrize("depth", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("include", [True, False])
def test_complete_graph_vs_oracle_all_endpoint_kinds(n, depth, include):
    """from/to inside or outside the graph (graph.rs:326-366): identical rows, identical order."""
    endpoints = {(n, n + 1), (0, n + 1), (n, 0), (0, 0), (0, 1), (1, 0), (n + 3, n + 3), (2, 2)}
    for fr, to in sorted(endpoints):
        exp_it = orc.CompleteGraphIter(n, fr, to, depth, include)
        declared = exp_it.remaining
        exp = exp_it.collect_array()
        g = CompleteGraph(n)
        it = g.all_paths(fr, to, depth, include_from_and_to=include)
        assert len(it) == min(declared, 2**63 - 1)
        if not exp_it.overflowed:
            assert declared == exp.shape[0]
        got = g.all_paths_array(fr, to, depth, include_from_and_to=include)
        assert got.shape == exp.shape, (n, depth, fr, to, got.shape, exp.shape)
        np.testing.assert_array_equal(got.astype(np.int64), exp)


@pytest.mark.parametrize("n,order", [(11, 1), (12, 3), (15, 4), (9, 5)])
@pytest.mark.parametrize("chunk_size", [1, 10, 23, 1000])
def test_chunks_iter(n, order, chunk_size):
    """differt/tests/geometry/test_utils.py:519-552."""
    full = generate_all_path_candidates(n, order)
    it = generate_all_path_candidates_chunks_iter(n, order, chunk_size)
    assert len(it) == -(-full.shape[0] // chunk_size)
    chunks = list(it)
    assert all(c.shape == (chunk_size, order) for c in chunks[:-1])
    assert 0 < chunks[-1].shape[0] <= chunk_size
    np.testing.assert_array_equal(np.concatenate(chunks), full)


def test_paths_iter_and_rank_window():
    """differt/tests/geometry/test_utils.py:493-516 + rank windows (the GPU enumerator's unit)."""
    full = generate_all_path_candidates(5, 4)
    it = generate_all_path_candidates_iter(5, 4)
    assert len(it) == full.shape[0] == 5 * 4**3
    rows = np.stack(list(it))
    np.testing.assert_array_equal(rows, full)
    win = CompleteGraph(5).all_paths_array(5, 6, 6, include_from_and_to=False, rank_lo=37, rank_hi=101)
    np.testing.assert_array_equal(win.astype(np.int32), full[37:101])


def test_count_formula_and_overflow():
    """differt-core/tests/geometry/test_graph.py:158-205; graph.rs:368-375."""
    for n, depth in [(10, 3), (100, 5), (1000, 4), (10000, 4), (200000, 4)]:
        it = CompleteGraph(n).all_paths(n, n + 1, depth, include_from_and_to=False)
        assert len(it) == n * (n - 1) ** (depth - 3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        it = CompleteGraph(10**6).all_paths(10**6, 10**6 + 1, 8)
        assert any("OverflowError" in str(x.message) for x in w)
    assert it._declared == 2**64 - 1
    # a window far into a huge (non-overflowing) space is still addressable
    n = 10000
    total = n * (n - 1) ** 2
    win = CompleteGraph(n).all_paths_array(n, n + 1, 5, include_from_and_to=False, rank_lo=total - 3)
    assert win.shape == (3, 3) and (win[:, 0] == 9999).all() and (win[-1] == [9999, 9998, 9999]).all() and (win[0] == [9999, 9998, 9996]).all()


def test_chunk_size_must_be_positive():
    with pytest.raises(ValueError):
        CompleteGraph(3).all_paths_array_chunks(3, 4, 4, chunk_size=0)


@pytest.mark.parametrize("n,depth", [(4, 3), (5, 4), (6, 5), (3, 2), (9, 2)])
def test_complete_graph_equals_digraph(n, depth):
    """graph.rs:1563-1577."""
    cg = CompleteGraph(n)
    exp = cg.all_paths_array(n, n + 1, depth)
    g = DiGraph.from_complete_graph(cg)
    f, t = g.insert_from_and_to_nodes(direct_path=True)
    assert (f, t) == (n, n + 1) and g.num_nodes == n + 2
    np.testing.assert_array_equal(g.all_paths_array(f, t, depth), exp)
    singles = list(g.all_paths(f, t, depth))
    assert len(singles) == exp.shape[0]
    chunks = list(g.all_paths_array_chunks(f, t, depth, chunk_size=7))
    np.testing.assert_array_equal(np.concatenate(chunks) if chunks else exp[:0], exp)


def test_digraph_vs_oracle_random_adjacency(rng):
    """Random adjacency + from/to adjacency vectors (hybrid-tracer style, SV:1013-1040)."""
    for n in (5, 9):
        adj = rng.random((n, n)) > 0.4
        np.fill_diagonal(adj, False)
        fa, ta = rng.random(n) > 0.3, rng.random(n) > 0.3
        g = DiGraph.from_adjacency_matrix(adj)
        f, t = g.insert_from_and_to_nodes(direct_path=False, from_adjacency=fa, to_adjacency=ta)
        og = orc.DiGraph.from_adjacency_matrix(adj)
        of, ot = og.insert_from_and_to_nodes(direct_path=False, from_adjacency=fa, to_adjacency=ta)
        assert (f, t) == (of, ot)
        for depth in (2, 3, 4, 5):
            for inc in (True, False):
                got = g.all_paths_array(f, t, depth, include_from_and_to=inc)
                exp = og.all_paths_array(of, ot, depth, include_from_and_to=inc)
                np.testing.assert_array_equal(got.astype(np.int64), exp)


@pytest.mark.parametrize("fast_mode", [True, False])
def test_mask_and_disconnect(fast_mode):
    """differt-core/tests/geometry/test_graph.py:120-142: masking == a smaller complete graph."""
    n, order = 7, 3
    mask = np.array([1, 0, 1, 1, 0, 1, 1], dtype=bool)
    g = DiGraph.from_complete_graph(CompleteGraph(n))
    f, t = g.insert_from_and_to_nodes()
    g.filter_by_mask(mask, fast_mode=fast_mode)
    got = g.all_paths_array(f, t, order + 2, include_from_and_to=False)
    active = np.flatnonzero(mask)
    exp = active[generate_all_path_candidates(len(active), order)]
    np.testing.assert_array_equal(got.astype(np.int64), exp)
    g2 = DiGraph.from_complete_graph(CompleteGraph(n))
    f, t = g2.insert_from_and_to_nodes()
    g2.disconnect_nodes(1, 4, fast_mode=fast_mode)
    np.testing.assert_array_equal(g2.all_paths_array(f, t, order + 2, include_from_and_to=False), got)
    with pytest.raises(ValueError):
        g2.filter_by_mask(np.ones(n + 3, dtype=bool))
    with pytest.raises(IndexError):
        g2.disconnect_nodes(n + 5)
