"""The smoothed-mode oracle (oracle/torch_ref.py) pinned on what the reference's own tests assert.

The reference has no numeric golden for the smoothed mode; its tests pin it through properties:
``smoothing_function`` == sigmoid with exact limits (differt/tests/test_utils.py:59-81) and "a large
smoothing factor matches no smoothing" for Moller-Trumbore (tests/geometry/test_utils.py:636-646),
any-triangle with a non-zero remainder tile (:701-715) and the same-side test
(tests/geometry/test_image_method.py:246-255).  Here the hard side of those equalities is the C
oracle, which IS pinned on the reference's golden vectors (tests/test_oracle_golden.py).
The autograd gradients of the restatement are checked against float64 central differences.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc
from oracle import torch_ref as tr

EPS, TOL, MINLEN = orc.DEFAULT_EPSILON, orc.DEFAULT_HIT_TOL, orc.DEFAULT_MIN_LEN


def _t(x, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype)


def test_smoothing_function(rng):
    """test_utils.py:59-81."""
    x = _t(rng.normal(size=(40, 1, 10)) * 1000.0)
    x[0, 0, 0] = 0.0
    sf = _t(rng.uniform(0, 100, size=(20, 1)))
    got = tr.smoothing_function(x, sf)
    np.testing.assert_allclose(got.numpy(), torch.sigmoid(x * sf).numpy(), rtol=2e-6, atol=1e-37)
    lim = tr.smoothing_function(_t([-1e8, 0.0, 1e8]), sf)
    np.testing.assert_allclose(lim.numpy(), np.broadcast_to(np.array([0.0, 0.5, 1.0], np.float32), lim.shape))
    big = tr.smoothing_function(x, 1e8)
    np.testing.assert_array_equal(big.numpy(), 0.5 * (np.sign(x.numpy()) + 1))
    assert not torch.isnan(tr.smoothing_function(_t([-np.inf, np.inf]), 3.0)).any()


@pytest.mark.parametrize("shape", [((10, 1, 3), (1, 15, 3, 3)), ((7, 3), (7, 3, 3)), ((3,), (3, 3))])
def test_large_smoothing_factor_matches_hard_mt(rng, shape):
    """tests/geometry/test_utils.py:609-646: t identical, hit > 0.5 == hard hit."""
    o = rng.uniform(-1, 1, shape[0]).astype(np.float32)
    d = rng.uniform(-1, 1, shape[0]).astype(np.float32)
    tv = rng.uniform(-1, 1, shape[1]).astype(np.float32)
    t_hard, hit_hard = orc.ray_intersect_triangle(o, d, tv)
    t, hit = tr.ray_intersect_triangle(_t(o), _t(d), _t(tv), epsilon=EPS, smoothing_factor=1e8)
    np.testing.assert_array_equal(t.numpy().view(np.uint32), t_hard.view(np.uint32))
    np.testing.assert_array_equal(hit.numpy() > 0.5, hit_hard)
    assert hit.dtype == torch.float32 and float(hit.min()) >= 0.0 and float(hit.max()) <= 1.0


def test_mt_known_answers_smoothed(goldens):
    """The reference's hand-written Moller-Trumbore cases (tests/geometry/test_utils.py:555-606) in
    the smoothed mode: t exact, confidence on the right side of 0.5 (steep and moderate slopes)."""
    g = goldens["ray_intersect_triangle_t_and_hit"]
    o = _t(g["ray_origin"])[None, None, :]
    d = _t(g["ray_directions"])[:, None, :]
    for sf in (50.0, 1e8):
        t, hit = tr.ray_intersect_triangle(o, d, _t(g["triangle_vertices"]), epsilon=EPS, smoothing_factor=sf)
        np.testing.assert_array_equal(t.numpy(), np.asarray(g["expected_t"], np.float32))
        # the rays cross the hypotenuse (u + v == 1 exactly): sigmoid(0) = 0.5, valid under the
        # `>= confidence_threshold` convention of TracedPaths (geometry/_paths.py:101-114)
        np.testing.assert_array_equal(hit.numpy() >= 0.5, np.asarray(g["expected_hit"]))
    g = goldens["ray_intersect_triangle_hits"]
    tri = _t([g["triangle"]])
    for case in g["cases"]:
        oo = _t(case["orig"])
        dd = _t(case["dest"]) - oo
        t, hit = tr.ray_intersect_triangle(oo, dd, tri, epsilon=EPS, smoothing_factor=1e8)
        assert bool(((t < 1.0) & (hit >= 0.5))[0]) == case["expected"]


@pytest.mark.parametrize("batch_size", [11, 512, None])
@pytest.mark.parametrize("with_active", [False, True])
def test_large_smoothing_factor_matches_hard_any(rng, batch_size, with_active):
    """tests/geometry/test_utils.py:649-715 (batch_size=11 leaves a remainder tile)."""
    o = rng.uniform(-1, 1, (30, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (30, 3)).astype(np.float32) * 3
    tv = rng.uniform(-1, 1, (30, 3, 3)).astype(np.float32)
    act = rng.random(30) > 0.3 if with_active else None
    hard = orc.ray_intersect_any_triangle(o, d, tv, act)
    got = tr.ray_intersect_any_triangle(_t(o), _t(d), _t(tv), None if act is None else torch.tensor(act),
                                        epsilon=EPS, hit_tol=TOL, smoothing_factor=1e8, batch_size=batch_size)
    np.testing.assert_array_equal(got.numpy() > 0.5, hard)
    assert hard.any() and not hard.all()
    none = tr.ray_intersect_any_triangle(_t(o), _t(d), _t(tv[:0]), epsilon=EPS, hit_tol=TOL, smoothing_factor=3.0)
    assert none.shape == (30,) and not none.any()


def test_any_triangle_clips_at_one(rng):
    """Many overlapping triangles in front of a ray: the tile sums are clipped to 1 (:1475-1476)."""
    tri = np.array([[-1, -1, 1], [3, -1, 1], [-1, 3, 1]], np.float32)
    tv = np.repeat(tri[None], 40, 0) + rng.normal(size=(40, 1, 3)).astype(np.float32) * 1e-3
    tv[:, :, 2] = np.linspace(0.2, 0.8, 40, dtype=np.float32)[:, None]
    got = tr.ray_intersect_any_triangle(_t([[0, 0, 0]]), _t([[0, 0, 1.0]]), _t(tv), epsilon=EPS, hit_tol=TOL,
                                        smoothing_factor=20.0, batch_size=7)
    assert float(got) == 1.0


def test_large_smoothing_factor_matches_hard_same_side(rng):
    """tests/geometry/test_image_method.py:222-255."""
    v = rng.normal(size=(10, 6, 3)).astype(np.float32)
    mv = rng.normal(size=(4, 3)).astype(np.float32)
    mn = rng.normal(size=(4, 3)).astype(np.float32)
    hard = orc.consecutive_vertices_are_on_same_side_of_mirror(v, mv, mn)
    got = tr.consecutive_vertices_are_on_same_side_of_mirror(_t(v), _t(mv), _t(mn), smoothing_factor=1e8)
    assert got.shape == (10, 4)
    np.testing.assert_array_equal(got.numpy() > 0.5, hard)


def _tb_case(two_buildings, goldens, order, assume_quads, mask):
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    n = Tr.shape[0] // 2 if assume_quads else Tr.shape[0]
    cand = orc.generate_all_path_candidates(n, order).astype(np.int64) * (2 if assume_quads else 1)
    return g, V, Tr, cand


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("assume_quads", [False, True])
@pytest.mark.parametrize("mesh_mask", [False, True])
def test_trace_large_smoothing_factor_matches_hard(two_buildings, goldens, order, assume_quads, mesh_mask):
    """differt/tests/geometry/test_scene.py:366-442 (xfail in the reference): with a steep slope the
    confidences threshold to the hard masks, except where a hard `==` / `>=` sits exactly on its
    boundary (sigmoid(0) = 0.5): those candidates are listed and must be boundary cases."""
    g, V, Tr, cand = _tb_case(two_buildings, goldens, order, assume_quads, None)
    rng = np.random.default_rng(5)
    mask = (rng.random(Tr.shape[0]) > 0.2) if mesh_mask else None
    if mask is not None and assume_quads:
        mask[1::2] = mask[0::2]
    hard = orc.trace_path_candidates(V, Tr, g["tx"], g["rx"], cand.astype(np.int32), mask=mask, assume_quads=assume_quads)
    full, m = tr.trace_smooth(_t(V), torch.tensor(Tr, dtype=torch.long), _t(g["tx"]).reshape(1, 3),
                              _t(g["rx"]).reshape(1, 3), torch.tensor(cand),
                              mask=None if mask is None else torch.tensor(mask), assume_quads=assume_quads,
                              epsilon=EPS, hit_tol=TOL, min_len=MINLEN, smoothing_factor=1e8)
    np.testing.assert_allclose(full.numpy(), hard["vertices"], rtol=1e-5, atol=1e-5)
    soft = m.numpy()
    # candidates whose image path is not finite (mirror parallel to the ray, IM:131-135) carry NaN
    # segments; jnp.min propagates them, so their confidence is NaN -- never >= the threshold
    nonfinite = np.isnan(soft)
    assert not (nonfinite & hard["mask"]).any()
    ok = soft[~nonfinite]
    assert ok.min() >= 0.0 and ok.max() <= 1.0
    np.testing.assert_array_equal(soft >= 0.5, hard["mask"])  # the golden valid path included
    assert (soft[hard["mask"]] == 1.0).all()
    if order == 2 and not mesh_mask:
        assert nonfinite.any()


def test_trace_padding_rows(two_buildings, goldens):
    g, V, Tr, cand = _tb_case(two_buildings, goldens, 2, False, None)
    cand = np.concatenate((cand[:5], np.full((3, 2), -1)))
    full, m = tr.trace_smooth(_t(V), torch.tensor(Tr, dtype=torch.long), _t(g["tx"]).reshape(1, 3),
                              _t(g["rx"]).reshape(1, 3), torch.tensor(cand), epsilon=EPS, hit_tol=TOL,
                              min_len=MINLEN, smoothing_factor=0.05)
    assert (m[..., 5:] == 0).all() and (full[..., 5:, :, :] == 0).all() and (m[..., :5] >= 0).all()


@pytest.mark.parametrize("assume_quads", [False, True])
def test_trace_smooth_autograd_vs_central_differences(assume_quads):
    """float64: torch.autograd over the restatement == central differences of the soft mask and of the
    path vertices, in tx, rx and the mesh vertices (the restatement is the gradient oracle of
    tests/test_smooth_gpu.py)."""
    rng = np.random.default_rng(2)
    V, Tr = orc.box_mesh(4.0, 3.0, 2.5, with_top=True)
    V = V.astype(np.float64) + rng.normal(size=V.shape) * 0.05
    Trl = torch.tensor(Tr, dtype=torch.long)
    n = 6 if assume_quads else 12
    cand = torch.tensor(orc.generate_all_path_candidates(n, 2).astype(np.int64) * (2 if assume_quads else 1))
    tx = np.array([[0.7, -0.4, 0.3]])
    rx = np.array([[-0.9, 0.5, -0.2], [0.2, 0.9, 0.6]])
    w = torch.tensor(rng.normal(size=(1, 2, cand.shape[0])))
    wv = torch.tensor(rng.normal(size=(1, 2, cand.shape[0], 4, 3)))

    def loss(Vt, txt, rxt):
        full, m = tr.trace_smooth(Vt, Trl, txt, rxt, cand, assume_quads=assume_quads, epsilon=EPS, hit_tol=TOL,
                                  min_len=MINLEN, smoothing_factor=4.0, batch_size=5)
        return (m * w).sum() + 1e-2 * (full * wv).sum()

    ins = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (V, tx, rx)]
    loss(*ins).backward()
    h = 1e-6
    for k, name in enumerate(("vertices", "tx", "rx")):
        base = [t.detach().clone() for t in ins]
        flat = base[k].reshape(-1)
        fd = np.zeros(flat.numel())
        for i in range(flat.numel()):
            old = float(flat[i])
            flat[i] = old + h
            up = float(loss(*base))
            flat[i] = old - h
            dn = float(loss(*base))
            flat[i] = old
            fd[i] = (up - dn) / (2 * h)
        got = ins[k].grad.numpy().reshape(-1)
        assert np.abs(got).max() > 1e-3, name
        np.testing.assert_allclose(got, fd, rtol=2e-5, atol=2e-7 * np.abs(fd).max(), err_msg=name)
