"""``Scene``: transmitters + receivers + mesh, and the ``trace_paths`` dispatch contract.

Mirrors ``differt/src/differt/geometry/_scene.py`` (``Scene`` fields, ``trace_paths`` :650-764,
grids :343-375).  Loaders, plotting, SBR launching and the MLM kernel are out of scope
(SURVEY.md section 8).
"""

from __future__ import annotations

import warnings
from collections.abc import Iterator
from dataclasses import dataclass, replace
from typing import Any, Literal

import numpy as np
import torch

from .._tensors import as_f32, as_i32
from ._mesh import Mesh
from ._paths import LaunchedPaths, TracedPaths
from ._solvers import (
    AbstractPathLauncher,
    AbstractPathTracer,
    ExhaustivePathTracer,
    HybridPathTracer,
    SBRPathLauncher,
)
from ._utils import SizedIterator

__all__ = ["Scene"]


@dataclass
class Scene:
    transmitters: torch.Tensor
    receivers: torch.Tensor
    mesh: Mesh

    def __post_init__(self):
        self.transmitters = as_f32(self.transmitters)
        self.receivers = as_f32(self.receivers)

    @property
    def num_transmitters(self) -> int:
        return int(np.prod(self.transmitters.shape[:-1], dtype=np.int64))

    @property
    def num_receivers(self) -> int:
        return int(np.prod(self.receivers.shape[:-1], dtype=np.int64))

    def set_assume_quads(self, flag: bool = True) -> "Scene":
        return replace(self, mesh=self.mesh.set_assume_quads(flag))

    def with_mesh(self, mesh: Mesh) -> "Scene":
        return replace(self, mesh=mesh)

    def with_transmitters(self, tx) -> "Scene":
        return replace(self, transmitters=tx)

    def with_receivers(self, rx) -> "Scene":
        return replace(self, receivers=rx)

    def _grid(self, m: int, n: int | None, height: float) -> torch.Tensor:
        """``[n, m, 3]`` grid over the mesh's horizontal bounding box (_scene.py:343-375)."""
        n = m if n is None else n
        v = self.mesh.vertices
        lo, hi = v.min(dim=0).values, v.max(dim=0).values
        x = torch.linspace(float(lo[0]), float(hi[0]), m, device=v.device)
        y = torch.linspace(float(lo[1]), float(hi[1]), n, device=v.device)
        xx, yy = torch.meshgrid(x, y, indexing="xy")
        return torch.stack((xx, yy, torch.full_like(xx, height)), dim=-1)

    def with_transmitters_grid(self, m: int = 50, n: int | None = 50, *, height: float = 1.5) -> "Scene":
        return replace(self, transmitters=self._grid(m, n, height))

    def with_receivers_grid(self, m: int = 50, n: int | None = 50, *, height: float = 1.5) -> "Scene":
        return replace(self, receivers=self._grid(m, n, height))

    def compute_paths(self, order: int | None = None, *, method: Literal["exhaustive", "sbr", "hybrid"] = "exhaustive",
                      chunk_size: int | None = None, num_rays: int = int(1e6), path_candidates=None,
                      epsilon=None, hit_tol=None, min_len=None, max_dist: float = 1e-3,
                      smoothing_factor=None, confidence_threshold: float = 0.5,
                      batch_size: int | None = 512, disconnect_inactive_triangles: bool = False):
        """Deprecated front end of the reference (_scene.py:1046-1248): dispatches to
        :meth:`trace_paths` / :meth:`launch_paths`."""
        warnings.warn("compute_paths is deprecated. Use trace_paths() or launch_paths() instead.",
                      DeprecationWarning, stacklevel=2)
        if method == "sbr":
            if order is None:
                raise ValueError("Argument 'order' is required.")
            return self.launch_paths(order, solver=SBRPathLauncher(num_rays=num_rays, max_dist=max_dist))
        if method == "hybrid":
            solver = HybridPathTracer(num_rays=num_rays, epsilon=epsilon, hit_tol=hit_tol, min_len=min_len,
                                      smoothing_factor=smoothing_factor,
                                      confidence_threshold=confidence_threshold, batch_size=batch_size,
                                      chunk_size=chunk_size)
        elif method == "exhaustive":
            solver = ExhaustivePathTracer(epsilon=epsilon, hit_tol=hit_tol, min_len=min_len,
                                          smoothing_factor=smoothing_factor,
                                          confidence_threshold=confidence_threshold, batch_size=batch_size,
                                          disconnect_inactive_triangles=disconnect_inactive_triangles,
                                          chunk_size=chunk_size)
        else:
            raise ValueError(f"Unknown method '{method}'.")
        return self.trace_paths(order, solver=solver, path_candidates=path_candidates)

    def launch_paths(self, order: int, *, solver: AbstractPathLauncher | Literal["sbr"] = "sbr",
                     **solver_kwargs: Any) -> LaunchedPaths:
        """Launch rays and bounce them ``order`` times (_scene.py:783-835); batch shape
        ``[*tx_batch, *rx_batch, num_rays]``."""
        if order is None:  # _scene.py:815-817
            raise ValueError("Argument 'order' is required.")
        if isinstance(solver, str):
            if solver != "sbr":
                raise ValueError(f"Unknown solver: {solver}")  # :823
            solver = SBRPathLauncher(**solver_kwargs)
        elif solver_kwargs:
            raise ValueError("solver_kwargs cannot be used when a solver instance is provided.")  # :826
        p = solver.launch_paths(self, order)
        batch = (*self.transmitters.shape[:-1], *self.receivers.shape[:-1], p.vertices.shape[2])
        return LaunchedPaths(
            p.vertices.reshape(*batch, order + 2, 3), p.objects.reshape(*batch, order + 2),
            p.masks.reshape(*batch, order + 1), p.interaction_types.reshape(*batch, order),
            p.confidence_threshold,
        )

    def trace_paths(
        self,
        order: int | None = None,
        *,
        solver: AbstractPathTracer | Literal["exhaustive", "hybrid", "beam"] = "exhaustive",
        path_candidates=None,
        chunk_size: int | None = None,
        compact: bool = False,
        **solver_kwargs: Any,
    ) -> TracedPaths | SizedIterator | Iterator[TracedPaths]:
        """Trace ray paths between all transmitters and receivers (_scene.py:650-764).

        Result batch shape: ``[*tx_batch, *rx_batch, num_candidates]`` (:762-764).  ``compact=True``
        (MI355X extension) returns only the valid paths, flattened, without building the dense
        arrays or the candidate table; for orders 1..3 of the exhaustive solver it runs through the pruned search
        (``drt_trace_paths_beam``) unless ``literal=True`` is among the solver arguments (``ExhaustivePathTracer.literal``:
        every candidate through the filter kernel) -- same paths, order, vertex bits and keys either way.  ``solver="beam"`` (MI355X extension, orders 0..3): the valid paths of the
        exhaustive solver -- same objects, order and vertex bits -- through the geometrically pruned search
        (``drt_trace_paths_beam``), always compact; ``solver_kwargs`` go to ``ExhaustivePathTracer`` except
        ``kappa`` / ``max_paths``, which go to the search.
        """
        if (order is None) == (path_candidates is None):  # _scene.py:692-695
            raise ValueError("You must specify one of 'order' or `path_candidates`, not both.")
        if isinstance(solver, str) and solver == "beam":
            if path_candidates is not None or chunk_size is not None:
                raise ValueError("solver='beam' enumerates the candidates itself: pass 'order' only.")
            beam_kwargs = {k: solver_kwargs.pop(k) for k in ("kappa", "max_paths") if k in solver_kwargs}
            return ExhaustivePathTracer(**solver_kwargs).trace_beam_pruned(self, order, **beam_kwargs)
        if isinstance(solver, str):
            if solver not in ("exhaustive", "hybrid"):
                raise ValueError(f"Unknown solver: {solver}")  # :702
            if chunk_size is not None:
                solver_kwargs = {**solver_kwargs, "chunk_size": chunk_size}
            cls = ExhaustivePathTracer if solver == "exhaustive" else HybridPathTracer
            solver = cls(**solver_kwargs)
        elif solver_kwargs or chunk_size is not None:
            raise ValueError("solver_kwargs cannot be used when a solver instance is provided.")  # :705

        hybrid = isinstance(solver, HybridPathTracer)
        if hybrid and getattr(solver, "smoothing_factor", None) is not None:  # :708-716
            warnings.warn("Argument 'smoothing' is currently ignored when using HybridPathTracer.",
                          UserWarning, stacklevel=2)  # the reference warns and still forwards it (SV:1173)
        if hybrid and order is None:  # :717-719
            raise ValueError("Argument 'order' is required when using HybridPathTracer.")
        if path_candidates is not None and getattr(solver, "chunk_size", None) is not None:  # :720-728
            warnings.warn("Argument 'chunk_size' is ignored when 'path_candidates' is provided.",
                          UserWarning, stacklevel=2)
            solver = replace(solver, chunk_size=None)

        tx_batch = tuple(self.transmitters.shape[:-1])
        rx_batch = tuple(self.receivers.shape[:-1])

        if path_candidates is not None:
            path_candidates = as_i32(path_candidates)
            if path_candidates.dim() != 2:
                raise ValueError("path_candidates must have shape [num_candidates, order]")
            if self.mesh.assume_quads:  # user-supplied ids are rounded down to the even triangle
                path_candidates = path_candidates - path_candidates % 2  # _scene.py:756-757

        if compact:
            if path_candidates is not None:
                return solver.trace_path_candidates_compact(self, path_candidates)
            return solver.trace_rank_range(self, order)  # hybrid: pruned product space, unranked on the GPU

        eff_chunk = getattr(solver, "chunk_size", None)
        if path_candidates is None and eff_chunk is not None:  # _scene.py:735-751
            chunks = solver.generate_path_candidates_chunks_iter(self, order, chunk_size=eff_chunk)

            def gen() -> Iterator[TracedPaths]:
                for cands, types in chunks:
                    p = solver.trace_path_candidates(self, cands, types)
                    yield p.reshape(*tx_batch, *rx_batch, cands.shape[0])

            if chunks.__len__() < 0:  # DiGraph chunk iterators do not know their length (size -1, SV:929-932)
                return gen()
            return SizedIterator(gen(), size=chunks.__len__)

        if path_candidates is None:
            cands, types = solver.generate_path_candidates(self, order)
        else:
            cands = path_candidates
            types = None
        paths = solver.trace_path_candidates(self, cands, types)
        return paths.reshape(*tx_batch, *rx_batch, cands.shape[0])
