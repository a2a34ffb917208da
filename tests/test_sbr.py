"""Shooting-and-bouncing rays ("next" row f3): fused HIP launch kernel vs the NumPy/C oracle
restatement of geometry/_solvers.py:279-491, 1179-1226, and the two-buildings SBR golden rows
(differt/tests/geometry/test_scene.py:116-260, method="sbr": the reference itself compares at
rtol = 1.0 there, after averaging the rays of each object sequence)."""

from __future__ import annotations

import numpy as np
import pytest

import oracle as orc


def test_oracle_sbr_two_buildings(goldens, two_buildings):
    """test_scene.py:205-238 on the oracle: SBR(max_dist=1e-1) finds the golden object sequences and
    bounce points (mean over the rays of a sequence, rtol 1.0 like the reference; in fact < 2 %)."""
    g = goldens["advanced_path_tracing_example"]
    V, Tr = two_buildings["vertices"], two_buildings["triangles"]
    ro, rd = orc.sbr_launch_rays(V, Tr, g["tx"], g["rx"], 200_000)
    for order in (0, 1, 2):
        out = orc.launch_paths(V, Tr, ro, rd, g["rx"], order, max_dist=1e-1)
        m = out["masks"][0, 0, :, order]
        assert m.any()
        if order:
            objs = out["triangles"][0][m]
            seqs = np.unique(objs - objs % 2, axis=0)
            exp = np.asarray(g["orders"][str(order)]["objects"], np.int32)[0, 1:-1]
            assert (seqs == (exp - exp % 2)).all(axis=1).any()
            sel = ((objs - objs % 2) == (exp - exp % 2)).all(axis=1)
            mean = out["vertices"][0][m][sel].mean(axis=0)
            np.testing.assert_allclose(mean, np.asarray(g["orders"][str(order)]["path_vertices"], np.float32)[0],
                                       rtol=0.05, atol=0.05)


gpu = pytest.mark.gpu


def _np(x):
    return x.detach().cpu().numpy()


@gpu
@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_gpu_launch_kernel_bit_exact(order):
    """Given identical rays, every output of the fused kernel equals the oracle: hit triangles,
    bounce points (bit for bit) and the per-order receiver masks."""
    import differt_amd.geometry as G
    from conftest import canyon_scene

    rng = np.random.default_rng(order)
    V, Tr = canyon_scene(rng, nextra=5)
    mask = rng.random(Tr.shape[0]) > 0.1
    tx = np.stack([rng.uniform(-20, 20, 2), rng.uniform(-3, 3, 2), rng.uniform(5, 15, 2)], -1).astype(np.float32)
    rx = np.stack([rng.uniform(-20, 20, 4), rng.uniform(-3, 3, 4), rng.uniform(1, 8, 4)], -1).astype(np.float32)
    R = 20_000
    rd = rng.normal(size=(2, R, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    ro = np.broadcast_to(tx[:, None], rd.shape).copy()
    exp = orc.launch_paths(V, Tr, ro, rd, rx, order, mask=mask, max_dist=4.0)

    class Fixed(G.AbstractPathLauncher):
        max_dist = 4.0

        def launch_rays(self, scene):
            import torch

            return torch.as_tensor(ro, device="cuda"), torch.as_tensor(rd, device="cuda")

    scene = G.Scene(tx, rx, G.Mesh(V, Tr, mask=mask))
    got = scene.launch_paths(order, solver=Fixed())
    assert tuple(got.masks.shape) == (2, 4, R, order + 1)
    np.testing.assert_array_equal(_np(got.masks), exp["masks"])
    np.testing.assert_array_equal(_np(got.objects)[:, 0, :, 1:-1], exp["triangles"])
    np.testing.assert_array_equal(_np(got.vertices)[:, 0, :, 1:-1].view(np.uint32), exp["vertices"].view(np.uint32))
    assert exp["masks"].sum() > 50 and (exp["triangles"] >= 0).mean() > 0.5 if order else True
    lower = got.get_paths(min(order, 1))
    assert lower.order == min(order, 1)
    with pytest.raises(ValueError):
        got.get_paths(order + 1)


@gpu
@pytest.mark.parametrize("order", [0, 1, 2])
def test_gpu_sbr_two_buildings_goldens(goldens, two_buildings, order):
    """test_scene.py:205-238 through Scene.launch_paths(solver="sbr")."""
    import differt_amd.geometry as G

    g = goldens["advanced_path_tracing_example"]
    scene = G.Scene(np.asarray(g["tx"], np.float32), np.asarray(g["rx"], np.float32),
                    G.Mesh(two_buildings["vertices"], two_buildings["triangles"]))
    got = scene.launch_paths(order, solver="sbr", num_rays=200_000, max_dist=1e-1)
    assert tuple(got.masks.shape) == (200_000, order + 1)
    mv, mo = _np(got.masked_vertices), _np(got.masked_objects)
    assert len(mv) > 0
    exp = g["orders"][str(order)]
    eo = np.asarray(exp["objects"], np.int32)[0]
    sel = ((mo - mo % 2) == (eo - eo % 2)).all(axis=1)
    assert sel.any()
    if order:
        np.testing.assert_allclose(mv[sel][:, 1:-1].mean(axis=0), np.asarray(exp["path_vertices"], np.float32)[0],
                                   rtol=0.05, atol=0.05)
    # lattice rays of the launcher == the oracle's up to trig ulps
    ro, rd = G.SBRPathLauncher(num_rays=5000).launch_rays(scene)
    ero, erd = orc.sbr_launch_rays(two_buildings["vertices"], two_buildings["triangles"], g["tx"], g["rx"], 5000)
    np.testing.assert_allclose(_np(rd), erd, atol=2e-5)
    np.testing.assert_array_equal(_np(ro), ero)


@gpu
@pytest.mark.parametrize("order", [1, 2, 3])
def test_gpu_launch_paths_vjp_vs_float64_autograd(order):
    """launch_paths is differentiable in the reference (_solvers.py:385-444: a lax.scan of plain JAX
    code around Mesh.first_triangle_hit_by_ray with its custom VJP, _mesh.py:258-344).  Here
    drt_launch_paths_vjp: gradients of a random linear functional of the bounce points w.r.t. the
    transmitter positions (ray origins), the ray directions and the mesh vertices, against float64
    autograd of the torch restatement of the same chain on the same hit triangles (<= 1e-5 rel)."""
    import torch

    import differt_amd.geometry as G
    from conftest import canyon_scene
    from oracle import torch_ref

    rng = np.random.default_rng(100 + order)
    V, Tr = canyon_scene(rng, nextra=4)
    ntx, R = 2, 600
    tx = np.stack([rng.uniform(-15, 15, ntx), rng.uniform(-2, 2, ntx), rng.uniform(5, 12, ntx)], -1).astype(np.float32)
    rx = np.zeros((1, 3), np.float32)
    rdn = rng.normal(size=(ntx, R, 3)).astype(np.float32)
    rdn /= np.linalg.norm(rdn, axis=-1, keepdims=True)
    W = rng.normal(size=(ntx, R, order, 3)).astype(np.float32)

    txg = torch.tensor(tx, device="cuda", requires_grad=True)
    rdg = torch.tensor(rdn, device="cuda", requires_grad=True)
    mesh = G.Mesh(V, Tr)
    vleaf = mesh.vertices.detach().clone().requires_grad_(True)  # the leaf (Mesh keeps a reshaped view)
    mesh = mesh.with_vertices(vleaf)

    class Fixed(G.AbstractPathLauncher):
        max_dist = 1.0

        def launch_rays(self, scene):
            return txg[:, None, :].expand(ntx, R, 3).contiguous(), rdg

    got = G.Scene(txg, rx, mesh).launch_paths(order, solver=Fixed())
    inner = got.vertices[:, 0, :, 1:-1]  # [ntx, R, order, 3]
    tris = _np(got.objects)[:, 0, :, 1:-1].astype(np.int64)
    # rays that leave the scene keep their last point: weight only complete chains so that the functional
    # is smooth (a miss has t = 0 and contributes through the reflection only, which the test also covers
    # through the first bounces of incomplete chains being excluded symmetrically on both sides)
    full = (tris >= 0).all(axis=-1)
    assert full.mean() > 0.3
    Wm = W * full[..., None, None]
    (inner * torch.tensor(Wm, device="cuda")).sum().backward()

    # float64 restatement on the same hit triangles
    V64 = torch.tensor(V.astype(np.float64), requires_grad=True)
    tx64 = torch.tensor(tx.astype(np.float64), requires_grad=True)
    rd64 = torch.tensor(rdn.astype(np.float64), requires_grad=True)
    T64 = torch.tensor(Tr.astype(np.int64))
    o = tx64[:, None, :].expand(ntx, R, 3).reshape(-1, 3)
    d = rd64.reshape(-1, 3)
    nrm = torch_ref.normals(V64, T64)
    pts = []
    for b in range(order):
        face = torch.tensor(tris[..., b].reshape(-1))
        t = torch_ref.differentiable_distance(V64, T64, o, d, face)
        inside = torch.isfinite(t)
        t = torch.where(inside, t, torch.zeros_like(t))
        o = o + t[:, None] * d
        n = nrm[torch.where(face >= 0, face, torch.full_like(face, Tr.shape[0] - 1))]
        d = d - 2.0 * (d * n).sum(-1, keepdim=True) * n
        pts.append(o)
    ref = torch.stack(pts, dim=1).reshape(ntx, R, order, 3)
    np.testing.assert_allclose(_np(inner)[full], ref.detach().numpy()[full], rtol=2e-4, atol=2e-4)
    (ref * torch.tensor(Wm.astype(np.float64))).sum().backward()
    for got_g, exp_g in ((txg.grad, tx64.grad), (rdg.grad, rd64.grad), (vleaf.grad, V64.grad)):
        e = exp_g.numpy()
        scale = float(np.abs(e).max()) + 1e-30
        assert float(np.abs(_np(got_g) - e).max()) <= 2e-5 * scale, (float(np.abs(_np(got_g) - e).max()), scale)
