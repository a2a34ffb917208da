"""Helper of the pruned-vs-exhaustive comparisons (not a test module): the one class of valid paths the pruned search
does not promise to keep.

A path is SHORT-SEGMENT DEGENERATE when two CONSECUTIVE REFLECTION points lie closer together than the pruning's own
error unit u = kappa * ulp(M) (M = largest coordinate magnitude of the scene).  The reference computes the direction of
that segment as the float32 difference of two points that coincide within the arithmetic's resolution: the ray its inside
test (geometry/_solvers.py:598-642 -> _utils.py:1263-1322) then runs on is rounding noise, its same-side test
(_solver_image_method.py:443-454) decides the sign of a distance below its own rounding error, and what the reference
accepts is no longer tied to the geometry of the mirrors (the reflection point may lie centimetres outside the triangle it
"hits": tests/golden/beam_cases/sub_ulp_segment_soup772.npz).  The reference's guard against such segments, `min_len`
(_solvers.py:684-693), is an ABSOLUTE 1.09 mm and does not follow the scale of the scene.  The exhaustive tracer reproduces
these artifacts bit for bit (it runs the reference's arithmetic on every candidate); a geometric pruning cannot be complete
for them without giving up pruning wherever a mirror's plane passes through the previous mirror."""

from __future__ import annotations

import numpy as np

KAPPA = 64.0


def ulp_of_scene(*arrays) -> float:
    m = max(float(np.abs(np.asarray(a, np.float64)).max()) for a in arrays if np.asarray(a).size)
    return float(np.spacing(np.float32(m)))


def short_segment_mask(vertices, ulp_m: float, kappa: float = KAPPA) -> np.ndarray:
    """vertices [N, k+2, 3] (tx, reflection points, rx) -> bool[N]: some segment BETWEEN two reflection points is shorter
    than kappa * ulp(M).  Orders 0 and 1 have no such segment."""
    v = np.asarray(vertices, np.float64)
    if v.ndim != 3 or v.shape[1] < 4:
        return np.zeros(v.shape[0] if v.ndim == 3 else 0, bool)
    seg = np.linalg.norm(v[:, 2:-1] - v[:, 1:-2], axis=-1)
    return (seg < kappa * ulp_m).any(axis=1)
