"""Path-candidate graphs with the public surface of ``differt_core.geometry``.

Mirrors ``differt-core/python/differt_core/_differt_core/geometry/graph.pyi:1-103`` (the PyO3
classes of ``differt-core/src/geometry/graph.rs``).  The native side is the host C++ of
``differt_amd/csrc/enumerate.cpp``: rows come from closed-form *unranking*, so any rank window can
be produced directly (and in parallel) instead of stepping an odometer.
"""

from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from .. import _lib

_LOG = logging.getLogger("differt_amd.geometry.graph")

__all__ = ["CompleteGraph", "DiGraph"]

_U64_MAX = (1 << 64) - 1


def _count(num_nodes: int, from_: int, to: int, depth: int) -> tuple[int, bool]:
    c, o = C.c_uint64(), C.c_int32()
    _lib.call("drt_complete_graph_count", num_nodes, from_, to, depth, C.byref(c), C.byref(o))
    return int(c.value), bool(o.value)


def _no_extra_positionals(name: str, allowed: int, extra: tuple) -> None:
    """PyO3's message for keyword-only parameters passed positionally (self is not counted)."""
    if extra:
        given = allowed + len(extra)
        raise TypeError(f"{name}() takes {allowed} positional arguments but {given} "
                        f"{'was' if given == 1 else 'were'} given")


class _CompleteGraphPathsIter:
    """``AllPathsFromCompleteGraphIter`` (graph.rs:286-491): sized iterator over single paths."""

    _BLOCK = 4096

    def __init__(self, num_nodes, from_, to, depth, include_from_and_to):
        self._args = (num_nodes, from_, to, depth, int(include_from_and_to))
        self._total, self._overflow = _count(num_nodes, from_, to, depth)
        if self._overflow:  # graph.rs:368-375
            # the reference logs a WARNING through pyo3-log (graph.rs:368-375), it does not raise / warn
            _LOG.warning("OverflowError: overflow occurred when computing the total number of paths, "
                         "defaulting to maximum value %d.", _U64_MAX)
            # the closed form overflowed; iterate up to the true count (usually astronomically large,
            # but tiny in degenerate cases such as num_nodes == 1)
            c, o = C.c_uint64(), C.c_int32()
            _lib.call("drt_complete_graph_count_exact", num_nodes, from_, to, depth, C.byref(c),
                      C.byref(o))
            self._declared = self._total
            self._total = int(c.value)
            self._exceeds = bool(o.value)
        else:
            self._declared, self._exceeds = self._total, False
        self._next = 0
        self._buf = None
        self._buf_lo = 0
        self.path_depth = depth if include_from_and_to else max(depth - 2, 0)

    def __iter__(self):
        return self

    def __len__(self) -> int:
        return min(self._declared - self._next, (1 << 63) - 1)

    def _fill(self, lo: int, hi: int) -> np.ndarray:
        out = np.zeros((hi - lo, self.path_depth), dtype=np.uint64)
        n, f, t, d, inc = self._args
        _lib.call("drt_complete_graph_fill_host", n, f, t, d, inc, lo, hi,
                  out.ctypes.data_as(C.c_void_p))
        return out

    def __next__(self) -> np.ndarray:
        if self._next >= self._total:
            raise StopIteration
        if self._buf is None or self._next >= self._buf_lo + len(self._buf):
            hi = min(self._next + self._BLOCK, self._total)
            self._buf, self._buf_lo = self._fill(self._next, hi), self._next
        row = self._buf[self._next - self._buf_lo]
        self._next += 1
        return row

    def count(self) -> int:
        """Remaining number of paths (consumes the iterator, like Rust's ``Iterator::count``)."""
        rem = self._total - self._next
        self._next = self._total
        return rem


class _CompleteGraphChunksIter:
    """``AllPathsFromCompleteGraphChunksIter`` (graph.rs:75-116, 257-276)."""

    def __init__(self, inner: _CompleteGraphPathsIter, chunk_size: int):
        self._it = inner
        self._chunk = chunk_size

    def __iter__(self):
        return self

    def __len__(self) -> int:
        return -(-len(self._it) // self._chunk)

    def __next__(self) -> np.ndarray:
        it = self._it
        if it._next >= it._total:
            raise StopIteration
        hi = min(it._next + self._chunk, it._total)
        out = it._fill(it._next, hi)
        it._next = hi
        return out


class CompleteGraph:
    """A complete graph on ``num_nodes`` nodes (graph.rs:127-277)."""

    def __init__(self, num_nodes: int) -> None:
        if num_nodes < 0:
            raise OverflowError("can't convert negative int to unsigned")
        self.num_nodes = int(num_nodes)

    def all_paths(self, from_: int, to: int, depth: int, *args, include_from_and_to: bool = True):
        """Iterator over all paths of ``depth`` nodes from ``from_`` to ``to`` (graph.rs:193-203);
        ``from_``/``to`` may lie outside the graph (``>= num_nodes``)."""
        _no_extra_positionals("CompleteGraph.all_paths", 3, args)
        return _CompleteGraphPathsIter(self.num_nodes, from_, to, depth, include_from_and_to)

    def all_paths_array(
        self, from_: int, to: int, depth: int, *, include_from_and_to: bool = True,
        rank_lo: int = 0, rank_hi: int | None = None,
    ) -> np.ndarray:
        """``UInt[ndarray, "num_paths path_depth"]`` of all paths (graph.rs:222-233).

        Extension: ``rank_lo``/``rank_hi`` select a window of the lexicographic order."""
        it = _CompleteGraphPathsIter(self.num_nodes, from_, to, depth, include_from_and_to)
        hi = it._total if rank_hi is None else min(int(rank_hi), it._total)
        if it._exceeds and rank_hi is None:
            raise MemoryError("the number of paths overflows 64 bits; pass a rank window")
        lo = min(int(rank_lo), hi)
        return it._fill(lo, hi)

    def all_paths_array_chunks(
        self, from_: int, to: int, depth: int, *, include_from_and_to: bool = True,
        chunk_size: int = 1000,
    ):
        """Iterator over chunks of at most ``chunk_size`` paths (graph.rs:257-276)."""
        if chunk_size <= 0:  # graph.rs:265 assert!
            raise ValueError("'chunk_size' must be strictly positive")
        return _CompleteGraphChunksIter(
            _CompleteGraphPathsIter(self.num_nodes, from_, to, depth, include_from_and_to),
            chunk_size,
        )


class _DiGraphIter:
    def __init__(self, graph: "DiGraph", from_, to, depth, include_from_and_to, chunk: int | None):
        self._graph = graph  # keep the native graph alive
        h = C.c_void_p()
        _lib.call("drt_digraph_iter_create", graph._h, from_, to, depth, int(include_from_and_to),
                  C.byref(h))
        self._h = h
        self._chunk = chunk
        self.path_depth = depth if include_from_and_to else max(depth - 2, 0)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _lib.load().drt_digraph_iter_destroy(self._h)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass
            self._h = None

    def __iter__(self):
        return self

    def _pull(self, max_rows: int) -> np.ndarray:
        out = np.zeros((max_rows, self.path_depth), dtype=np.uint64)
        rows = C.c_uint64()
        _lib.call("drt_digraph_iter_next_chunk", self._h, max_rows,
                  out.ctypes.data_as(C.c_void_p), C.byref(rows))
        return out[: rows.value]

    def __next__(self) -> np.ndarray:
        got = self._pull(1 if self._chunk is None else self._chunk)
        if len(got) == 0:
            raise StopIteration
        return got[0] if self._chunk is None else got


class DiGraph:
    """A directed graph stored as sorted adjacency lists (graph.rs:594-1010)."""

    def __init__(self, handle) -> None:
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _lib.load().drt_digraph_destroy(self._h)
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass
            self._h = None

    @classmethod
    def empty(cls, num_nodes: int) -> "DiGraph":
        return cls.from_adjacency_matrix(np.zeros((num_nodes, num_nodes), dtype=bool))

    @classmethod
    def from_adjacency_matrix(cls, adjacency_matrix) -> "DiGraph":
        """graph.rs:616-633."""
        m = np.ascontiguousarray(np.asarray(adjacency_matrix, dtype=np.uint8))
        if m.ndim != 2 or m.shape[0] != m.shape[1]:
            raise ValueError("'adjacency_matrix' must be square")
        h = C.c_void_p()
        _lib.call("drt_digraph_from_adjacency_matrix", m.ctypes.data_as(C.c_void_p), m.shape[0],
                  C.byref(h))
        return cls(h)

    @classmethod
    def from_complete_graph(cls, graph: CompleteGraph) -> "DiGraph":
        """graph.rs:1012-1024."""
        h = C.c_void_p()
        _lib.call("drt_digraph_from_complete_graph", graph.num_nodes, C.byref(h))
        return cls(h)

    @property
    def num_nodes(self) -> int:
        return int(_lib.load().drt_digraph_num_nodes(self._h))

    def insert_from_and_to_nodes(
        self, *args, direct_path: bool = True, from_adjacency=None, to_adjacency=None
    ) -> tuple[int, int]:
        """graph.rs:636-691 (all parameters keyword-only, graph.pyi)."""
        _no_extra_positionals("DiGraph.insert_from_and_to_nodes", 0, args)
        n = self.num_nodes
        fa = ta = None
        if from_adjacency is not None:
            fa = np.ascontiguousarray(np.asarray(from_adjacency, dtype=np.uint8))
            if fa.shape != (n,):
                raise ValueError("'from_adjacency' must have exactly 'num_nodes' elements")
        if to_adjacency is not None:
            ta = np.ascontiguousarray(np.asarray(to_adjacency, dtype=np.uint8))
            if ta.shape != (n,):
                raise ValueError("'to_adjacency' must have exactly 'num_nodes' elements")
        f, t = C.c_uint64(), C.c_uint64()
        _lib.call(
            "drt_digraph_insert_from_and_to_nodes", self._h, int(direct_path),
            None if fa is None else fa.ctypes.data_as(C.c_void_p),
            None if ta is None else ta.ctypes.data_as(C.c_void_p), C.byref(f), C.byref(t),
        )
        return int(f.value), int(t.value)

    def disconnect_nodes(self, *nodes: int, fast_mode: bool = True) -> None:
        """graph.rs:833-850."""
        arr = np.asarray(nodes, dtype=np.uint64)
        try:
            _lib.call("drt_digraph_disconnect_nodes", self._h, arr.ctypes.data_as(C.c_void_p),
                      len(arr), int(fast_mode))
        except ValueError as e:
            raise IndexError(str(e)) from None

    def filter_by_mask(self, mask, fast_mode: bool = True) -> None:
        """graph.rs:879-910 (a mask shorter than the graph leaves the remaining nodes connected)."""
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        if len(m) > self.num_nodes:  # graph.rs:886-893
            raise ValueError(f"'mask' length ({len(m)}) must be smaller than or equal to the number of nodes "
                             f"in the graph ({self.num_nodes})")
        _lib.call("drt_digraph_filter_by_mask", self._h, m.ctypes.data_as(C.c_void_p), len(m),
                  int(fast_mode))

    def all_paths(self, from_: int, to: int, depth: int, *args, include_from_and_to: bool = True):
        """graph.rs:912-925 (unsized iterator)."""
        _no_extra_positionals("DiGraph.all_paths", 3, args)
        return _DiGraphIter(self, from_, to, depth, include_from_and_to, None)

    def all_paths_array(self, from_: int, to: int, depth: int, *, include_from_and_to: bool = True):
        """graph.rs:927-943."""
        it = _DiGraphIter(self, from_, to, depth, include_from_and_to, 1 << 16)
        chunks = list(it)
        if not chunks:
            return np.zeros((0, it.path_depth), dtype=np.uint64)
        return np.concatenate(chunks, axis=0)

    def all_paths_array_chunks(
        self, from_: int, to: int, depth: int, *, include_from_and_to: bool = True,
        chunk_size: int = 1000,
    ):
        """graph.rs:945-975."""
        if chunk_size <= 0:
            raise ValueError("'chunk_size' must be strictly positive")
        return _DiGraphIter(self, from_, to, depth, include_from_and_to, chunk_size)
