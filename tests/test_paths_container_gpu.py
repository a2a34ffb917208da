"""Container methods of ``TracedPaths`` / ``LaunchedPaths`` that group, deduplicate and reduce paths
(reference geometry/_paths.py:21-74, 152-252, 331-479, 600-688), bit-exact integer work on the GPU
(``drt_row_cell_ids``, csrc/groups.hip).  Mirrors differt/tests/geometry/test_paths.py:25-36, 184-450
and the ``group_by_objects`` docstring example (geometry/_paths.py:391-418).
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


def _np(x):
    return x.detach().cpu().numpy()


def _random_paths(G, rng, path_length, *batch, num_objects=30, with_mask=True):
    dev = "cuda"
    v = torch.tensor(rng.random((*batch, path_length, 3)), dtype=torch.float32, device=dev)
    o = torch.tensor(rng.integers(0, num_objects, (*batch, path_length)), dtype=torch.int32, device=dev)
    m = torch.tensor(rng.random(batch) > 0.5 if with_mask else np.ones(batch, bool), device=dev)
    it = torch.zeros((*batch, path_length - 2), dtype=torch.int32, device=dev)
    return G.TracedPaths(v, o, m, it)


def test_merge_cell_ids(G):
    """test_paths.py:25-35."""
    got = G.merge_cell_ids([4, 0, 2, 0, 4], [1, 3, 7, 3, 1])
    assert _np(got).tolist() == [0, 1, 2, 1, 0]
    a = np.arange(12).reshape(3, 4) % 3
    assert tuple(G.merge_cell_ids(a, a.T.reshape(3, 4) % 2).shape) == (3, 4)


@pytest.mark.parametrize(("n", "width", "num_values"), [(1, 3, 2), (7, 1, 2), (1000, 3, 3), (5000, 6, 2),
                                                       (200_000, 4, 6), (100_000, 40, 2), (33, 0, 1)])
def test_cell_ids_vs_oracle(G, rng, n, width, num_values):
    from differt_amd.geometry._paths import _cell_ids

    rows = rng.integers(-1, num_values, (n, width)).astype(np.int32)
    got = _np(_cell_ids(torch.tensor(rows, device="cuda")))
    np.testing.assert_array_equal(got, orc.cell_ids(rows))
    assert (got <= np.arange(n)).all() and (got[got] == got).all()


def test_group_by_objects(G, rng):
    """geometry/_paths.py:391-418 (docstring known answer) and test_paths.py:339-368."""
    objects = np.array([[[1, 1, 0], [0, 0, 1], [1, 0, 1], [1, 0, 0], [1, 1, 1], [1, 1, 1]],
                        [[1, 0, 0], [1, 1, 1], [0, 0, 1], [1, 1, 0], [0, 0, 1], [1, 0, 0]]], np.int32)
    p = G.TracedPaths(torch.zeros((2, 6, 3, 3), device="cuda"), torch.tensor(objects, device="cuda"),
                      torch.ones((2, 6), dtype=torch.bool, device="cuda"))
    assert _np(p.group_by_objects()).tolist() == [[0, 1, 2, 3, 4, 4], [3, 4, 1, 0, 1, 3]]
    for path_length in (3, 5):
        for batch in ((), (1,), (1, 2, 3, 4), (50, 40)):
            paths = _random_paths(G, rng, path_length, *batch, num_objects=2, with_mask=False)
            got = paths.group_by_objects()
            assert tuple(got.shape) == batch
            obj = _np(paths.objects).reshape(-1, path_length)
            np.testing.assert_array_equal(_np(got).reshape(-1), orc.cell_ids(obj))


def test_multipath_cells(G, rng):
    """test_paths.py:370-411 (bool and float masks)."""
    paths = _random_paths(G, rng, 3, 6, 2, num_objects=1)
    m = np.array([[1, 0], [1, 1], [1, 0], [0, 0], [0, 1], [0, 0]], bool)
    paths.mask = torch.tensor(m, device="cuda")
    assert _np(paths.multipath_cells()).tolist() == [0, 1, 0, 3, 4, 3]
    paths.mask = torch.tensor([[0.8, 0.2], [0.9, 0.7], [0.6, 0.1], [0.2, 0.3], [0.4, 0.8], [0.1, 0.0]], device="cuda")
    assert _np(paths.multipath_cells()).tolist() == [0, 1, 0, 3, 4, 3]
    big = _random_paths(G, rng, 3, 4, 5, 64)
    got = big.multipath_cells(axis=-1)
    assert tuple(got.shape) == (4, 5)
    np.testing.assert_array_equal(_np(got).reshape(-1), orc.cell_ids(_np(big.mask).reshape(20, 64)))
    assert tuple(big.multipath_cells(axis=0).shape) == (5, 64)


def test_mask_duplicate_objects(G, rng):
    """test_paths.py:203-305."""
    mesh = G.Mesh.box()
    cand = np.array([[0, 1, 2], [1, 0, 2], [0, 1, 2], [0, 1, 2], [2, 3, 4], [1, 0, 2]], np.int32)
    scene = G.Scene(rng.normal(size=3).astype(np.float32), rng.normal(size=3).astype(np.float32), mesh)
    paths = scene.trace_paths(path_candidates=cand)
    paths.mask = torch.ones_like(paths.mask)
    got = paths.mask_duplicate_objects()
    assert int(got.num_valid_paths) == 3 and _np(got.mask).tolist() == [True, True, False, False, True, False]
    with pytest.raises(ValueError, match="The provided axis -2 is out-of-bounds for batch of dimensions 1!"):
        paths.mask_duplicate_objects(axis=-2)
    scene = scene.with_transmitters_grid(2, 1).with_receivers_grid(4, 3)
    paths = scene.trace_paths(path_candidates=cand)
    assert tuple(paths.mask.shape) == (1, 2, 3, 4, 6)
    paths.mask = torch.ones_like(paths.mask)
    got = paths.mask_duplicate_objects()
    assert tuple(got.mask.shape) == (1, 2, 3, 4, 6) and int(got.num_valid_paths) == 3 * 24
    sw = G.TracedPaths(paths.vertices.swapaxes(0, -3), paths.objects.swapaxes(0, -2), paths.mask.swapaxes(0, -1),
                       paths.interaction_types.swapaxes(0, -2))
    got = sw.mask_duplicate_objects(axis=0)
    assert tuple(got.mask.shape) == (6, 2, 3, 4, 1) and int(got.num_valid_paths) == 3 * 24
    sw.mask = torch.ones((6, 2, 3, 4, 1), dtype=torch.float32, device="cuda")
    assert int(sw.mask_duplicate_objects(axis=0).num_valid_paths) == 3 * 24
    # an already-masked first occurrence is not replaced by a later duplicate (mask * first-occurrence)
    paths1 = scene.trace_paths(path_candidates=cand)
    keep = paths1.mask_duplicate_objects()
    assert bool((keep.mask <= paths1.mask).all())


@pytest.mark.parametrize(("batch", "axis", "ok"), [((), None, True), ((), 0, False), ((1,), 0, True), ((1, 2, 1), None, True),
                                                   ((1, 2, 1), (0, 2), True), ((1, 2, 1), -1, True), ((1, 2), 2, False)])
def test_squeeze(G, rng, batch, axis, ok):
    """test_paths.py:150-201."""
    paths = _random_paths(G, rng, 10, *batch)
    if not ok:
        with pytest.raises(ValueError):
            paths.squeeze(axis=axis)
        return
    got = paths.squeeze(axis=axis)
    exp = np.squeeze(_np(paths.mask), axis=axis)
    assert tuple(got.mask.shape) == exp.shape and got.shape == exp.shape
    assert tuple(got.vertices.shape) == (*exp.shape, 10, 3) and tuple(got.interaction_types.shape) == (*exp.shape, 8)


def test_iter_and_reduce(G, rng):
    """test_paths.py:413-450."""
    import differt_amd.em._utils as emu

    paths = _random_paths(G, rng, 6, 3, 2, num_objects=20)
    n = 0
    for p in paths:
        n += 1
        assert isinstance(p, G.TracedPaths) and int(p.num_valid_paths) == 1 and tuple(p.vertices.shape) == (6, 3)
    assert n == int(paths.num_valid_paths)
    for batch, axis, shape in (((), None, ()), ((10,), None, ()), ((5, 20), -1, (5,)), ((15, 20), (0, 1), ())):
        paths = _random_paths(G, rng, 4, *batch, num_objects=3)
        lengths = emu.path_length(paths.vertices)
        exp = torch.where(paths.mask, lengths, torch.zeros_like(lengths))
        exp = exp.sum() if axis is None else exp.sum(dim=axis)
        got = paths.reduce(emu.path_length, axis=axis)
        assert tuple(got.shape) == shape and torch.equal(got, exp)
    soft = _random_paths(G, rng, 4, 7)
    soft.mask = torch.rand(7, device="cuda")
    np.testing.assert_allclose(_np(soft.reduce(emu.path_length)), _np((emu.path_length(soft.vertices) * soft.mask).sum()))


def test_launched_paths_methods(G, rng):
    """test_paths.py:459-560 (reshape / squeeze / iteration / masked of LaunchedPaths)."""
    from conftest import canyon_scene

    V, Tr = canyon_scene(rng)
    scene = G.Scene([[-15.0, 1.0, 8.0]], [[12.0, -2.0, 3.0], [5.0, 2.0, 2.0]], G.Mesh(V, Tr))
    lp = scene.launch_paths(order=2, num_rays=20_000, max_dist=1.0)
    assert lp.shape == (1, 2, lp.vertices.shape[2])
    flat = lp.reshape(-1)
    assert flat.shape == (2 * lp.vertices.shape[2],) and torch.equal(flat.masks.reshape(lp.masks.shape), lp.masks)
    sq = lp.squeeze(0)
    assert sq.shape == lp.shape[1:] and torch.equal(sq.masked_objects, lp.masked_objects)
    m = lp.masked()
    assert isinstance(m, G.TracedPaths) and m.vertices.shape[0] == int(lp.mask.sum())
    assert sum(1 for _ in lp) == int(lp.mask.sum())
    dd = lp.get_paths(2).mask_duplicate_objects()
    assert int(dd.num_valid_paths) <= int(lp.mask.sum())
    if int(lp.mask.sum()):
        assert int(dd.num_valid_paths) >= 1
