"""Keys of the default compact tracer (round 6): the pruned search returns packed keys, `ExhaustivePathTracer._rank_keyed` turns
them into the rank keys of the exhaustive enumeration -- `(tx * num_rx + rx) * total + rank`, rank = position of (m_1 .. m_k) in
the lexicographic enumeration of the complete graph (differt-core/src/geometry/graph.rs:301-397).  Pure index arithmetic: checked
here on the CPU against the oracle's enumerator, for plain meshes, quads and disconnected (masked) nodes."""

from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle as orc


def _paths(objs):
    from differt_amd.geometry._paths import TracedPaths

    o = torch.as_tensor(objs, dtype=torch.int32)
    k = o.shape[1] - 2
    return TracedPaths(torch.zeros((o.shape[0], k + 2, 3)), o, torch.ones(o.shape[0], dtype=torch.bool),
                       torch.zeros((o.shape[0], k), dtype=torch.int32), 0.5, torch.zeros(o.shape[0], dtype=torch.int64))


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("mode", ["plain", "quads", "disconnected"])
def test_rank_keys_equal_the_position_in_the_enumeration(order, mode):
    from differt_amd.geometry._solvers import ExhaustivePathTracer

    n_prim, ntx, nrx = 6, 2, 3
    if mode == "disconnected":
        active = np.array([1, 0, 1, 1, 0, 1], bool)
        node_map = torch.as_tensor(np.flatnonzero(active).astype(np.int32))
        n = int(active.sum())
    else:
        node_map, n = None, n_prim
    cand = orc.generate_all_path_candidates(n, order).astype(np.int64)        # node indices, lexicographic
    ids = cand if node_map is None else node_map.numpy().astype(np.int64)[cand]  # primitive ids
    if mode == "quads":
        ids = ids * 2                                                          # objects carry the even triangle id of a quad
    total = cand.shape[0]
    rows = []
    for it in range(ntx):
        for ir in range(nrx):
            rows.append(np.column_stack((np.full(total, it), ids, np.full(total, ir))))
    objs = np.concatenate(rows)
    mesh = SimpleNamespace(assume_quads=(mode == "quads"), num_primitives=n_prim)
    scene = SimpleNamespace(mesh=mesh, receivers=torch.zeros((nrx, 3)))
    got = ExhaustivePathTracer()._rank_keyed(scene, _paths(objs), order, n, node_map, total)
    assert torch.equal(got.keys, torch.arange(ntx * nrx * total, dtype=torch.int64))
    assert torch.equal(got.objects, torch.as_tensor(objs, dtype=torch.int32))
