"""``TracedPaths``: the output container of the tracer (reference geometry/_paths.py:77-328)."""

from __future__ import annotations

from dataclasses import dataclass, replace

import torch

__all__ = ["LaunchedPaths", "TracedPaths"]


@dataclass
class TracedPaths:
    """Ray paths with the layout of the reference (geometry/_paths.py:85-110):
    ``vertices f32[*batch, path_length, 3]``, ``objects i32[*batch, path_length]``,
    ``mask bool[*batch]``, ``interaction_types i32[*batch, order]``."""

    vertices: torch.Tensor
    objects: torch.Tensor
    mask: torch.Tensor
    interaction_types: torch.Tensor | None = None
    confidence_threshold: float = 0.5
    keys: torch.Tensor | None = None  # compact mode: flat [tx, rx, candidate] index of each path

    @property
    def shape(self) -> tuple[int, ...]:
        return tuple(self.objects.shape[:-1])

    @property
    def path_length(self) -> int:
        return self.objects.shape[-1]

    @property
    def order(self) -> int:
        return self.path_length - 2

    def _bool_mask(self) -> torch.Tensor:
        m = self.mask
        return m if m.dtype == torch.bool else (m >= self.confidence_threshold)

    @property
    def num_valid_paths(self) -> torch.Tensor:
        """geometry/_paths.py:264-272."""
        return self._bool_mask().sum()

    @property
    def masked_vertices(self) -> torch.Tensor:
        """geometry/_paths.py:274-283: valid paths, batch flattened, row-major order."""
        return self.vertices.reshape(-1, self.path_length, 3)[self._bool_mask().reshape(-1)]

    @property
    def masked_objects(self) -> torch.Tensor:
        """geometry/_paths.py:285-297."""
        return self.objects.reshape(-1, self.path_length)[self._bool_mask().reshape(-1)]

    def reshape(self, *batch: int) -> "TracedPaths":
        """geometry/_paths.py:123-150."""
        it = self.interaction_types
        objects = self.objects.reshape(*batch, self.path_length)
        batch = tuple(objects.shape[:-1])  # resolves a -1 (ambiguous for the empty order-0 arrays)
        return replace(
            self,
            vertices=self.vertices.reshape(*batch, self.path_length, 3),
            objects=objects,
            mask=self.mask.reshape(*batch),
            interaction_types=None if it is None else it.reshape(*batch, self.order),
        )

    def masked(self) -> "TracedPaths":
        """geometry/_paths.py:299-328: flattened, valid paths only."""
        p = self.reshape(-1)
        m = p._bool_mask()
        it = p.interaction_types
        return replace(
            p,
            vertices=p.vertices[m],
            objects=p.objects[m],
            mask=m[m],
            interaction_types=None if it is None else it[m],
        )


@dataclass
class LaunchedPaths:
    """Paths produced by ray launching (reference geometry/_paths.py:513-): one mask per path order
    (``masks[..., k]`` = the order-k path of that ray reaches the receiver), lower orders included."""

    vertices: torch.Tensor           # [*batch, order+2, 3]
    objects: torch.Tensor            # [*batch, order+2]
    masks: torch.Tensor              # [*batch, order+1]
    interaction_types: torch.Tensor | None = None
    confidence_threshold: float = 0.5

    @property
    def shape(self) -> tuple[int, ...]:
        return tuple(self.vertices.shape[:-2])

    @property
    def path_length(self) -> int:
        return self.objects.shape[-1]

    @property
    def order(self) -> int:
        return self.path_length - 2

    @property
    def mask(self) -> torch.Tensor:
        """Highest-order mask (geometry/_paths.py ``LaunchedPaths.mask``)."""
        return self.masks[..., -1]

    def get_paths(self, order: int) -> TracedPaths:
        """``TracedPaths`` of the given order (first ``order`` bounces + receiver)."""
        if order < 0 or order > self.order:
            raise ValueError(
                f"Paths order must be strictly between 0 and {self.order} (incl.), but you provided {order}."
            )
        v = torch.cat((self.vertices[..., : order + 1, :], self.vertices[..., -1:, :]), dim=-2)
        o = torch.cat((self.objects[..., : order + 1], self.objects[..., -1:]), dim=-1)
        it = None if self.interaction_types is None else self.interaction_types[..., :order]
        return TracedPaths(v, o, self.masks[..., order], it, self.confidence_threshold)

    @property
    def masked_vertices(self) -> torch.Tensor:
        return self.get_paths(self.order).masked_vertices

    @property
    def masked_objects(self) -> torch.Tensor:
        return self.get_paths(self.order).masked_objects
