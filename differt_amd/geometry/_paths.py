"""``TracedPaths``: the output container of the tracer (reference geometry/_paths.py:77-328)."""

from __future__ import annotations

from collections.abc import Callable, Iterator, Sequence
from dataclasses import dataclass, replace

import torch

from .. import _lib
from .._tensors import ptr, stream

__all__ = ["LaunchedPaths", "TracedPaths", "merge_cell_ids"]


def _cell_ids(rows: torch.Tensor) -> torch.Tensor:
    """``out[r]`` = smallest index of a row equal to ``rows[r]`` (reference ``_cell_ids``,
    geometry/_paths.py:21-38), on the GPU: ``drt_row_cell_ids`` (csrc/groups.hip)."""
    rows = rows.to(torch.int32).contiguous()
    n, w = rows.shape
    out = torch.empty(n, dtype=torch.int32, device=rows.device)
    if n:
        nbytes = _lib.load().drt_row_cell_ids_workspace_size(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=rows.device)
        _lib.call("drt_row_cell_ids", ptr(rows), n, w, ptr(out), ptr(ws), nbytes, stream())
    return out


def merge_cell_ids(cell_ids_a, cell_ids_b) -> torch.Tensor:
    """geometry/_paths.py:41-74: cells of the pairs ``(a[i], b[i])``."""
    from .._tensors import as_i32

    a, b = as_i32(cell_ids_a), as_i32(cell_ids_b)
    return _cell_ids(torch.stack((a, b), dim=-1).reshape(-1, 2)).reshape(a.shape)


def _squeeze_axes(ndim: int, axis) -> tuple[int, ...] | None:
    """Axis validation of geometry/_paths.py:165-181."""
    if axis is not None and ndim == 0:
        raise ValueError("Cannot squeeze a 0-dimensional batch!")
    if isinstance(axis, int):
        axis = (axis,)
    if isinstance(axis, Sequence):
        axis = tuple(a + ndim if a < 0 else a for a in axis)
        if any(ax >= ndim or ax < 0 for ax in axis):
            raise ValueError("One of the provided axes is out-of-bounds!")
    return axis


def _squeeze(t: torch.Tensor, axis, ndim: int) -> torch.Tensor:
    if axis is None:
        axis = tuple(i for i in range(ndim) if t.shape[i] == 1)
    for ax in axis:
        if t.shape[ax] != 1:
            raise ValueError(f"cannot select an axis to squeeze out which has size not equal to one (axis {ax})")
    return t.reshape(tuple(s for i, s in enumerate(t.shape) if i not in axis))


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

    def _attach_valid_count(self, counters: torch.Tensor) -> None:
        """Device-side counters of the dense tracer call that produced ``mask`` (``[0]`` rows that passed the geometric
        checks, ``[1]`` those the occlusion stage cleared): ``num_valid_paths`` reads them instead of reducing the mask,
        for as long as ``mask`` is the tensor the kernel wrote (same storage, same version, same element count -- reshapes
        and squeezes keep it, any in-place edit, slice or replacement falls back to the reduction)."""
        self.__dict__["_valid_count"] = (self.mask.data_ptr(), self.mask._version, self.mask.numel(), counters)

    @property
    def num_valid_paths(self) -> torch.Tensor:
        """geometry/_paths.py:264-272 (one reduction kernel over the 1-byte mask; `.sum()` would first widen it to int64;
        straight from the tracer's device-side counters when this object is what the dense tracer returned)."""
        vc = self.__dict__.get("_valid_count")
        m = self.mask
        if vc is not None and m.dtype == torch.bool and vc[0] == m.data_ptr() and vc[1] == m._version and vc[2] == m.numel():
            return vc[3][0] - vc[3][1]
        return torch.count_nonzero(self._bool_mask())

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
        out = replace(
            self,
            vertices=self.vertices.reshape(*batch, self.path_length, 3),
            objects=objects,
            mask=self.mask.reshape(*batch),
            interaction_types=None if it is None else it.reshape(*batch, self.order),
        )
        return self._carry_valid_count(out)

    def _carry_valid_count(self, out: "TracedPaths") -> "TracedPaths":
        vc = self.__dict__.get("_valid_count")
        if vc is not None:  # (checked against the new mask's storage / version / size when it is read)
            out.__dict__["_valid_count"] = vc
        return out

    def squeeze(self, axis: int | Sequence[int] | None = None) -> "TracedPaths":
        """geometry/_paths.py:152-194."""
        ndim = self.vertices.dim() - 2
        axis = _squeeze_axes(ndim, axis)
        it = self.interaction_types
        return self._carry_valid_count(replace(
            self, vertices=_squeeze(self.vertices, axis, ndim), objects=_squeeze(self.objects, axis, ndim),
            mask=_squeeze(self.mask, axis, ndim), interaction_types=None if it is None else _squeeze(it, axis, ndim)))

    def mask_duplicate_objects(self, axis: int = -1) -> "TracedPaths":
        """geometry/_paths.py:196-252: along ``axis`` only the FIRST path of every set with identical
        ``objects`` keeps its mask (``jnp.unique(..., return_index=True)`` keeps first occurrences)."""
        batch = self.shape
        ndim = len(batch)
        if ndim == 0 or not -ndim <= axis < ndim:
            raise ValueError(f"The provided axis {axis} is out-of-bounds for batch of dimensions {ndim}!")
        ax = axis % ndim
        obj = self.objects.movedim(ax, -2)                      # [*rest, size, L]
        size, L = obj.shape[-2], obj.shape[-1]
        flat = obj.reshape(-1, size, L)
        slice_id = torch.arange(flat.shape[0], dtype=torch.int32, device=flat.device)[:, None, None]
        rows = torch.cat((slice_id.expand(-1, size, 1), flat.to(torch.int32)), dim=-1).reshape(-1, L + 1)
        ids = _cell_ids(rows)
        first = ids == torch.arange(rows.shape[0], dtype=torch.int32, device=rows.device)
        keep = first.reshape(*obj.shape[:-1]).movedim(-1, ax)
        return replace(self, mask=self.mask * keep)

    def multipath_cells(self, axis: int = -1) -> torch.Tensor:
        """geometry/_paths.py:331-376: equal ids for batch entries whose validity pattern along
        ``axis`` is the same."""
        m = self._bool_mask().movedim(axis, -1)
        partial, last = m.shape[:-1], m.shape[-1]
        return _cell_ids(m.reshape(-1, last).to(torch.int32)).reshape(partial)

    def group_by_objects(self) -> torch.Tensor:
        """geometry/_paths.py:378-421."""
        return _cell_ids(self.objects.reshape(-1, self.path_length)).reshape(self.shape)

    def __iter__(self) -> Iterator["TracedPaths"]:
        """geometry/_paths.py:423-446: the valid paths, one at a time."""
        m = self.masked()
        for i in range(m.vertices.shape[0]):
            it = None if m.interaction_types is None else m.interaction_types[i]
            yield TracedPaths(m.vertices[i], m.objects[i], torch.ones((), dtype=torch.bool, device=m.mask.device),
                              it, m.confidence_threshold)

    def reduce(self, fun: Callable[[torch.Tensor], torch.Tensor], axis=None) -> torch.Tensor:
        """geometry/_paths.py:448-479: sum of ``fun(vertices)`` over the valid paths (weighted by the
        confidences for float masks)."""
        val = fun(self.vertices)
        if self.mask.dtype != torch.bool:
            out = val * self.mask
        else:
            out = torch.where(self.mask, val, torch.zeros_like(val))
        return out.sum() if axis is None else out.sum(dim=axis)

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

    def reshape(self, *batch: int) -> "LaunchedPaths":
        """geometry/_paths.py:600-627."""
        objects = self.objects.reshape(*batch, self.path_length)
        batch = tuple(objects.shape[:-1])
        it = self.interaction_types
        return replace(self, vertices=self.vertices.reshape(*batch, self.path_length, 3), objects=objects,
                       masks=self.masks.reshape(*batch, self.masks.shape[-1]),
                       interaction_types=None if it is None else it.reshape(*batch, self.order))

    def squeeze(self, axis: int | Sequence[int] | None = None) -> "LaunchedPaths":
        """geometry/_paths.py:629-670."""
        ndim = self.vertices.dim() - 2
        axis = _squeeze_axes(ndim, axis)
        it = self.interaction_types
        return replace(self, vertices=_squeeze(self.vertices, axis, ndim), objects=_squeeze(self.objects, axis, ndim),
                       masks=_squeeze(self.masks, axis, ndim),
                       interaction_types=None if it is None else _squeeze(it, axis, ndim))

    def __iter__(self) -> Iterator[TracedPaths]:
        """geometry/_paths.py:672-679."""
        yield from self.get_paths(self.order)

    def masked(self) -> TracedPaths:
        """geometry/_paths.py:681-688."""
        return self.get_paths(self.order).masked()

    @property
    def masked_vertices(self) -> torch.Tensor:
        return self.get_paths(self.order).masked_vertices

    @property
    def masked_objects(self) -> torch.Tensor:
        return self.get_paths(self.order).masked_objects
