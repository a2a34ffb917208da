"""Host-side helpers of the beam-pruned tracer that need no GPU: Morton clustering of receivers."""

from __future__ import annotations

import pytest
import torch

from differt_amd.geometry._solvers import _morton_order, _receiver_clusters


@pytest.mark.parametrize("R", [1, 2, 63, 64, 65, 257, 1024])
def test_receiver_clusters_cover_their_members(R):
    g = torch.Generator().manual_seed(R)
    rx = (torch.rand(R, 3, generator=g) - 0.5) * torch.tensor([1400.0, 1400.0, 30.0])
    rs, idx, boxes = _receiver_clusters(rx)
    assert rs.dtype == torch.float32 and idx.dtype == torch.int32 and boxes.shape == ((R + 63) // 64, 6)
    assert sorted(idx.tolist()) == list(range(R)) and torch.equal(rs, rx[idx.long()])
    for c in range(boxes.shape[0]):
        m = rs[c * 64:(c + 1) * 64].double()
        assert bool(((m - boxes[c, :3].double()).abs() <= boxes[c, 3:].double()).all())


def test_flat_grid_clusters_into_tiles():
    """A 32 x 32 street-level grid (BASELINE configs[4]) must split into 16 compact 8 x 8 tiles: one scale for all
    axes in the Morton code, otherwise the (zero) height range would eat a third of the bits."""
    xs = torch.linspace(-700.0, 700.0, 32)
    gx, gy = torch.meshgrid(xs, xs, indexing="ij")
    rx = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.full((1024,), 1.5)], dim=1)
    _, _, boxes = _receiver_clusters(rx)
    assert boxes.shape[0] == 16
    tile = 7 * float(xs[1] - xs[0]) / 2  # half extent of an 8 x 8 tile
    assert float(boxes[:, 3:5].max()) <= tile * 1.001 + 1e-3 and float(boxes[:, 5].max()) < 1e-3


def test_morton_order_is_a_permutation_and_local():
    g = torch.Generator().manual_seed(3)
    p = torch.rand(4096, 3, generator=g, dtype=torch.float64) * 100
    perm = _morton_order(p)
    assert sorted(perm.tolist()) == list(range(4096))
    q = p[perm]
    # consecutive points along the curve are much closer than random pairs
    step = (q[1:] - q[:-1]).norm(dim=-1).mean()
    rand = (p[1:] - p[:-1]).norm(dim=-1).mean()
    assert float(step) < 0.35 * float(rand)
