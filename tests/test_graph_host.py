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
    it = CompleteGraph(10**6).all_paths(10**6, 10**6 + 1, 8)  # logs a WARNING, see the mirrored test below
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


# ------------------------------------------------------------------------------------------
# differt-core/tests/geometry/test_graph.py mirrored one to one (host classes over the C ABI)
# ------------------------------------------------------------------------------------------
class TestDiGraphReference:
    def test_insert_from_and_to_nodes(self):
        """test_graph.py:14-31."""
        graph = DiGraph.from_complete_graph(CompleteGraph(5))
        assert graph.insert_from_and_to_nodes() == (5, 6)
        assert graph.insert_from_and_to_nodes(direct_path=True) == (7, 8)
        assert graph.insert_from_and_to_nodes(direct_path=False) == (9, 10)
        with pytest.raises(TypeError, match="takes 0 positional arguments but 1 was given"):
            graph.insert_from_and_to_nodes(True)

    @pytest.mark.parametrize("fast_mode", [True, False])
    def test_disconnect_nodes(self, fast_mode):
        """test_graph.py:33-49."""
        graph = DiGraph.from_complete_graph(CompleteGraph(6))
        from_, to = graph.insert_from_and_to_nodes()
        nodes = (0, 1, 2, 5)
        graph.disconnect_nodes(*nodes, fast_mode=fast_mode)
        for depth in range(3):
            for path in graph.all_paths(from_, to, depth + 2, include_from_and_to=False):
                assert not set(nodes) & set(int(x) for x in path)

    @pytest.mark.parametrize("fast_mode", [True, False])
    def test_filter_by_mask(self, fast_mode):
        """test_graph.py:51-80."""
        graph = DiGraph.from_complete_graph(CompleteGraph(8))
        from_, to = graph.insert_from_and_to_nodes()
        mask = np.array([True, False, True, False, True, False, True, False])
        before = len(list(graph.all_paths(from_, to, 3, include_from_and_to=False)))
        graph.filter_by_mask(mask, fast_mode=fast_mode)
        after = list(graph.all_paths(from_, to, 3, include_from_and_to=False))
        assert all(not {1, 3, 5, 7} & set(int(x) for x in p) for p in after)
        assert 0 < len(after) < before

    @pytest.mark.parametrize("fast_mode", [True, False])
    def test_filter_by_mask_all_disconnected(self, fast_mode):
        """test_graph.py:82-92."""
        graph = DiGraph.from_complete_graph(CompleteGraph(4))
        from_, to = graph.insert_from_and_to_nodes(direct_path=False)
        graph.filter_by_mask(np.zeros(4, bool), fast_mode=fast_mode)
        assert len(list(graph.all_paths(from_, to, 4, include_from_and_to=False))) == 0

    def test_filter_by_mask_wrong_size(self):
        """test_graph.py:94-118."""
        import re

        graph = DiGraph.from_complete_graph(CompleteGraph(5))
        graph.filter_by_mask(np.array([True, False, True]), fast_mode=True)  # a smaller mask is fine
        graph = DiGraph.from_complete_graph(CompleteGraph(5))
        with pytest.raises(ValueError, match=re.escape(
                "'mask' length (6) must be smaller than or equal to the number of nodes in the graph (5)")):
            graph.filter_by_mask(np.array([True, False, True, False, False, False]), fast_mode=True)

    @pytest.mark.parametrize("fast_mode", [True, False])
    def test_disconnect_nodes_equivalence(self, fast_mode):
        """test_graph.py:120-142."""
        complete = CompleteGraph(3)
        di = DiGraph.from_complete_graph(CompleteGraph(6))
        from_, to = di.insert_from_and_to_nodes()
        di.disconnect_nodes(3, 4, 5, fast_mode=fast_mode)
        for depth in range(3):
            a = complete.all_paths_array(from_, to, depth + 2, include_from_and_to=False)
            b = di.all_paths_array(from_, to, depth + 2, include_from_and_to=False)
            np.testing.assert_equal(np.asarray(a), np.asarray(b))

    def test_from_graph_and_keyword_only(self):
        """test_graph.py:144-158."""
        graph = CompleteGraph(10)
        assert isinstance(graph, CompleteGraph)
        graph = DiGraph.from_complete_graph(graph)
        assert isinstance(graph, DiGraph)
        with pytest.raises(TypeError, match="takes 3 positional arguments but 4 were given"):
            DiGraph.from_complete_graph(CompleteGraph(5)).all_paths(0, 1, 0, True)

    @pytest.mark.parametrize(("num_nodes", "depth"), [(15, 2), (25, 3), (11, 4)])
    def test_all_paths_count_from_complete_graph(self, num_nodes, depth):
        """test_graph.py:160-178."""
        graph = CompleteGraph(num_nodes)
        from_, to = num_nodes, num_nodes + 1
        n = sum(1 for _ in graph.all_paths(from_, to, depth + 2, include_from_and_to=False))
        assert n == num_nodes * (num_nodes - 1) ** (depth - 1)
        assert graph.all_paths_array(from_, to, depth + 2, include_from_and_to=False).shape == (n, depth)

    @pytest.mark.parametrize(("num_nodes", "depth"), [(10, 1), (50, 2), (10, 3)])
    def test_all_paths_count_from_di_graph(self, num_nodes, depth):
        """test_graph.py:180-199."""
        graph = DiGraph.from_complete_graph(CompleteGraph(num_nodes))
        from_, to = graph.insert_from_and_to_nodes()
        n = sum(1 for _ in graph.all_paths(from_, to, depth + 2, include_from_and_to=False))
        assert n == num_nodes * (num_nodes - 1) ** (depth - 1)
        assert graph.all_paths_array(from_, to, depth + 2, include_from_and_to=False).shape == (n, depth)

    @pytest.mark.parametrize(("num_nodes", "depth"), [(10, 100), (50_000, 10)])
    def test_all_paths_count_overflow_is_logged(self, num_nodes, depth, caplog):
        """test_graph.py:201-226."""
        import logging

        graph = CompleteGraph(num_nodes)
        caplog.clear()
        with caplog.at_level(logging.WARNING):
            graph.all_paths(num_nodes, num_nodes + 1, depth + 2, include_from_and_to=False)
        assert "OverflowError: overflow occurred when computing the total number of paths" in caplog.text
