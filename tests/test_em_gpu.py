"""GPU parity of the EM post-processing (csrc/em.hip through the C ABI) vs oracle/em_ref.py.

Floating-point row: tolerance 1e-5 relative (+ small absolute terms stated per quantity); the exact
known answers of the reference's tests are asserted on the GPU results as well.
Mirrors differt/tests/em/test_utils.py, test_fresnel.py and tests/plugins/test_deepmimo.py:28-105.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle as orc
from oracle import em_ref as emo

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def G():
    import differt_amd.geometry as g

    return g


@pytest.fixture(scope="module")
def EM():
    import differt_amd.em as em

    return em


def _np(x):
    return x.detach().cpu().numpy()


def test_path_delay(EM, rng):
    """test_utils.py:33-59."""
    for shape in ((10, 3), (20, 10, 3), (1, 3), (0, 3), (7, 5, 4, 3)):
        path = (rng.normal(size=shape) * 30).astype(np.float32)
        got = EM.path_delay(path)
        np.testing.assert_array_equal(_np(got).view(np.uint32), emo.path_delay(path).view(np.uint32))
    np.testing.assert_allclose(_np(EM.length_to_delay([1.0, 2.0, 4.0], speed=2.0)), [0.5, 1.0, 2.0])
    np.testing.assert_allclose(_np(EM.path_delay([[1.0, 0, 0], [1.0, 1.0, 0]])) * EM.c, 1.0, rtol=1e-6)


def test_sp_directions(EM, rng):
    """test_utils.py:62-88 + random parity (incl. normal incidence -> perpendicular_vector)."""
    cos, sin = np.cos(np.pi / 6), np.sin(np.pi / 6)
    k_i = np.array([[cos, -sin, 0.0], [0.0, -1.0, 0.0]], np.float32)
    k_r = np.array([[cos, +sin, 0.0], [0.0, +1.0, 0.0]], np.float32)
    n = np.array([[0.0, 1.0, 0.0], [0.0, 1.0, 0.0]], np.float32)
    (eis, eip), (ers, erp) = EM.sp_directions(k_i, k_r, n)
    assert torch.equal(eis, ers)
    np.testing.assert_allclose(_np(eis), [[0, 0, 1], [1, 0, 0]], atol=1e-6)
    np.testing.assert_allclose(_np(eip), [[sin, cos, 0], [0, 0, -1]], atol=1e-6)
    np.testing.assert_allclose(_np(erp), [[-sin, cos, 0], [0, 0, 1]], atol=1e-6)
    ki = emo.normalize(rng.normal(size=(64, 5, 3)))[0]
    nn = emo.normalize(rng.normal(size=(5, 3)))[0]
    ki[0] = -nn  # exact normal incidence on the first row
    kr = (ki - 2 * (ki * nn).sum(-1, keepdims=True) * nn).astype(np.float32)
    got = EM.sp_directions(ki, kr, nn)
    exp = emo.sp_directions(ki, kr, nn)
    for g, e in zip((*got[0], *got[1]), (*exp[0], *exp[1])):
        np.testing.assert_allclose(_np(g), e, rtol=RTOL, atol=1e-6)


def test_sp_rotation_matrix(EM, rng):
    """test_utils.py:91-123."""
    e_i_s, e_i_p = np.array([1.0, 0, 0], np.float32), np.array([0, 1.0, 0], np.float32)
    got = _np(EM.sp_rotation_matrix(e_i_s, e_i_p, [0, 1.0, 0], [-1.0, 0, 0]))
    np.testing.assert_allclose(got, [[0, 1], [-1, 0]], atol=1e-7)
    got = _np(EM.sp_rotation_matrix(e_i_s, e_i_p, e_i_s, -e_i_p))
    np.testing.assert_allclose(got, [[1, 0], [0, -1]])
    a, b, c, d = (rng.normal(size=(9, 4, 3)).astype(np.float32) for _ in range(4))
    np.testing.assert_array_equal(_np(EM.sp_rotation_matrix(a, b, c, d)), emo.sp_rotation_matrix(a, b, c, d))


def test_fspl(EM, rng):
    """test_utils.py:126-137."""
    d = rng.uniform(1, 100, (30, 1)).astype(np.float32)
    f = rng.uniform(0.1e9, 10e9, (1, 50)).astype(np.float32)
    got, got_db = _np(EM.fspl(d, f)), _np(EM.fspl(d, f, dB=True))
    np.testing.assert_allclose(10 * np.log10(got), got_db, rtol=1e-5)
    np.testing.assert_allclose(got_db, 20 * np.log10(d) + 20 * np.log10(f) - 147.55, rtol=2e-4)


def test_fresnel(EM, rng):
    """test_fresnel.py:32-92."""
    n_r = (rng.uniform(0.01, 2.0, 100) / rng.uniform(0.01, 2.0, 100)).astype(np.complex64)[:, None]
    theta = np.linspace(0, np.pi / 2, 50)
    ct = np.cos(theta).astype(np.float32)[None, :]
    (r_s, r_p), (t_s, t_p) = EM.fresnel_coefficients(n_r, ct)
    (er_s, er_p), (et_s, et_p) = emo.fresnel_coefficients(n_r, ct)
    for g, e in ((r_s, er_s), (r_p, er_p), (t_s, et_s), (t_p, et_p)):
        assert g.dtype == torch.complex64
        np.testing.assert_allclose(_np(g), e, rtol=RTOL, atol=2e-6)
    a, b = EM.reflection_coefficients(n_r, ct)
    assert torch.equal(a, r_s) and torch.equal(b, r_p)
    np.testing.assert_allclose(_np(t_s), _np(r_s) + 1, atol=1e-6)
    np.testing.assert_allclose(n_r * _np(t_p), _np(r_p) + 1, atol=2e-6)
    # lossy (complex) indices
    n_c = (rng.uniform(1, 3, 200) - 1j * rng.uniform(0, 1, 200)).astype(np.complex64)
    c2 = rng.uniform(0, 1, 200).astype(np.float32)
    got, exp = EM.fresnel_coefficients(n_c, c2), emo.fresnel_coefficients(n_c, c2)
    for g, e in zip((*got[0], *got[1]), (*exp[0], *exp[1])):
        np.testing.assert_allclose(_np(g), e, rtol=RTOL, atol=2e-6)
    # known answers: normal incidence, Brewster, refractive index of glass
    r_s, r_p = EM.reflection_coefficients(1.5, 1.0)
    assert complex(r_s) == -complex(r_p)
    _, r_p = EM.reflection_coefficients(1.5, float(np.cos(np.arctan(np.float32(1.5)))))
    assert abs(complex(r_p)) < 1e-7
    np.testing.assert_allclose(float(EM.refractive_index(EM.materials["Glass"].relative_permittivity(1e9))),
                               2.503997, rtol=1e-6)
    assert float(EM.refractive_index(EM.materials["itu_vacuum"].relative_permittivity(1e9))) == 1.0


def _box_scene(G, rng, ntx=2, nrx=3):
    mesh = G.Mesh.box(4.0, 3.0, 2.5, with_top=True)
    tx = rng.uniform(-0.9, 0.9, (ntx, 3)).astype(np.float32)
    rx = rng.uniform(-0.9, 0.9, (nrx, 3)).astype(np.float32)
    return G.Scene(tx, rx, mesh)


CHANNEL_KEYS = ("power", "phase", "length", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el")


def _assert_channel_close(got: dict, exp: dict, valid):
    np.testing.assert_array_equal(_np(got["length"]).view(np.uint32), exp["length"].view(np.uint32))
    np.testing.assert_array_equal(_np(got["delay"]).view(np.uint32), exp["delay"].view(np.uint32))
    a, e = _np(got["a"]), exp["a"]
    # cross-polar coefficients are small differences of co-polar sized terms: the error scale of a
    # path is its free-space amplitude lambda / (4 pi s) (times 1e-2), not the coefficient itself
    friis = (emo.c / 2.4e9) / (4 * np.pi * np.maximum(exp["length"], 1e-9))
    scale = np.abs(e) + 1e-2 * friis
    assert (np.abs(a - e)[valid] <= 2e-5 * scale[valid]).all(), float((np.abs(a - e) / scale)[valid].max())
    co = np.abs(e) > 1e-2 * friis  # well-conditioned (co-polar sized) coefficients
    ok = valid & co
    np.testing.assert_allclose(_np(got["power"])[ok], exp["power"][ok], rtol=RTOL, atol=2e-4)  # dB
    dphi = (_np(got["phase"]) - exp["phase"] + 180.0) % 360.0 - 180.0
    assert (np.abs(dphi)[ok] <= 2e-3).all()  # degrees
    for k in ("aoa_az", "aoa_el", "aod_az", "aod_el"):
        d = (_np(got[k]) - exp[k] + 180.0) % 360.0 - 180.0
        assert (np.abs(d)[valid] <= 1e-3).all(), k


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("polarization", ["V", "H", ("V", "H"), (0.3, -0.5, 0.8), ((0.0, 1.0, 0.0), "V")])
def test_paths_channel_vs_oracle(G, rng, order, polarization):
    from differt_amd.plugins import deepmimo

    scene = _box_scene(G, rng)
    mesh = scene.mesh.set_face_materials(np.arange(12) % 3)
    paths = scene.trace_paths(order)
    if paths.mask.shape[-1] > 400:
        keep = np.sort(rng.choice(paths.mask.shape[-1], 400, replace=False))
        paths = scene.trace_paths(path_candidates=_np(paths.objects)[0, 0, keep, 1:-1])
    f = 2.4e9
    n_c = emo.complex_refractive_index([5.24, 6.27, 2.0], [0.09, 0.012, 0.0], f)
    th = np.array([-1.0, 0.02, 0.3], np.float32)  # half space, thin lossy slab, lossless slab
    got = deepmimo.paths_channel(paths, mesh, np.stack((n_c.real, n_c.imag), -1), th, f, polarization)
    normals = orc.mesh_normals(orc.triangle_vertices(_np(mesh.vertices), _np(mesh.triangles)))
    exp = emo.channel(_np(paths.vertices), _np(paths.objects), normals, np.arange(12) % 3, n_c, th, f, polarization)
    valid = _np(paths.mask)
    assert valid.any() and got["a"].dtype == torch.complex64 and tuple(got["a"].shape) == valid.shape
    _assert_channel_close(got, exp, valid)


def test_complex_refractive_index_host():
    from differt_amd.plugins import deepmimo
    import differt_amd.em as em

    mats = {"A": em.Material.from_itu_properties("A", 5.24, 0.0, 0.0462, 0.7822),
            "B": em.Material("B", lambda f: (3.0, 0.5), thickness=0.1)}
    n, th = deepmimo.material_tables(("A", "B"), mats, 3.5e9)
    exp = emo.complex_refractive_index([5.24, 3.0], [0.0462 * 3.5**0.7822, 0.5], 3.5e9)
    np.testing.assert_allclose(n[:, 0] + 1j * n[:, 1], exp, rtol=1e-6)
    np.testing.assert_array_equal(th, np.array([-1.0, 0.1], np.float32))


def test_export(G, rng):
    """tests/plugins/test_deepmimo.py:28-105."""
    from differt_amd.plugins import deepmimo

    scene = _box_scene(G, rng, ntx=2, nrx=4)
    scene = G.Scene(scene.transmitters.reshape(1, 2, 3), scene.receivers, scene.mesh)
    with pytest.raises(ValueError, match="Scene must contain information about face materials"):
        deepmimo.export(paths=scene.trace_paths(order=0), scene=scene, frequency=2.4e9)
    scene = scene.with_mesh(scene.mesh.set_materials("itu_concrete"))
    f = 2.4e9
    for order in (0, 1, 2):
        paths = scene.trace_paths(order=order)
        dm = deepmimo.export(paths=paths, scene=scene, frequency=f)
        assert (dm.num_tx, dm.num_rx, dm.num_paths) == (2, 4, paths.vertices.shape[-3])
        it = scene.trace_paths(order=order, solver=G.ExhaustivePathTracer(chunk_size=100))
        dm2 = deepmimo.export(paths=it, scene=scene, radio_materials=deepmimo.materials, frequency=f)
        assert dm2.num_paths == dm.num_paths
        assert torch.equal(dm2.power, dm.power) and torch.equal(dm2.mask, dm.mask)
    dm = deepmimo.export(paths=(scene.trace_paths(order=k) for k in (0, 1, 2)), scene=scene, frequency=f,
                         include_primitives=True)
    assert dm.num_paths == 1 + 12 + 132 and dm.primitives.shape == (2, 4, 145, 2) and dm.inter_pos.shape[-2:] == (2, 3)
    assert int(dm.primitives[0, 0, 0, 0]) == -1 and int(dm.inter[0, 0, 0, 0]) == -1  # LOS rows are padded
    assert deepmimo.export(paths=scene.trace_paths(order=1), scene=scene, frequency=f).primitives is None
    nd = dm.numpy()
    assert all(isinstance(v, np.ndarray) for v in nd.asdict().values() if v is not None)
    assert len(dm.asdict()) == 13
    # against the oracle, whole container, valid paths only
    p2 = scene.trace_paths(order=2)
    mesh = scene.mesh
    normals = orc.mesh_normals(orc.triangle_vertices(_np(mesh.vertices), _np(mesh.triangles)))
    n_c = emo.complex_refractive_index([5.24], [0.0462 * 2.4**0.7822], f)
    exp = emo.channel(_np(p2.vertices), _np(p2.objects), normals, np.zeros(12, int), n_c, [-1.0], f)
    sl = slice(13, 145)
    valid = _np(p2.mask).reshape(2, 4, -1)
    assert valid.any()
    np.testing.assert_allclose(_np(dm.power)[..., sl][valid], exp["power"].reshape(2, 4, -1)[valid], rtol=RTOL, atol=2e-4)
    np.testing.assert_array_equal(_np(dm.delay)[..., sl], exp["delay"].reshape(2, 4, -1))
    np.testing.assert_array_equal(_np(dm.mask)[..., sl], valid)
    # soft masks are exported as confidences
    soft = deepmimo.export(paths=scene.trace_paths(order=1, solver=G.ExhaustivePathTracer(smoothing_factor=50.0)),
                           scene=scene, frequency=f)
    assert soft.mask.dtype == torch.float32
    with pytest.raises(ValueError, match="Unknown polarization"):
        deepmimo.export(paths=p2, scene=scene, frequency=f, polarization="X")


def test_mesh_materials(G):
    """geometry/_mesh.py:1930-2003."""
    mesh = G.Mesh.box(with_top=True)
    assert mesh.face_materials is None and mesh.material_names == ()
    m1 = mesh.set_materials("itu_concrete")
    assert m1.material_names == ("itu_concrete",) and _np(m1.face_materials).tolist() == [0] * 12
    names = ["a", "b"] * 6
    m2 = m1.set_materials(*names)
    assert m2.material_names == ("itu_concrete", "a", "b") and _np(m2.face_materials).tolist() == [1, 2] * 6
    q = mesh.set_assume_quads().set_materials(*"abcdef")
    assert _np(q.face_materials).tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]
    with pytest.raises(ValueError, match="Expected either 1, or 12 names"):
        mesh.set_materials("a", "b")
    with pytest.raises(ValueError, match="Expected either 1, 12, or 6 names"):
        mesh.set_assume_quads().set_materials("a", "b")
    assert _np(mesh.set_face_materials(3).face_materials).tolist() == [3] * 12
    # append (_mesh.py:1571-1575): names merged, the second mesh renumbered, -1 where a mesh has no materials
    a = G.Mesh.box().set_materials("concrete")
    b = G.Mesh.box().set_materials(*(["glass", "concrete"] * 5))
    c = a + b
    assert c.material_names == ("concrete", "glass")
    assert _np(c.face_materials).tolist() == [0] * 10 + [1, 0] * 5
    d = G.Mesh.box() + a
    assert _np(d.face_materials).tolist() == [-1] * 10 + [0] * 10 and d.material_names == ("concrete",)
    assert (G.Mesh.box() + G.Mesh.box()).face_materials is None


# ------------------------------------------------------------------ gradients ----
def _oracle64(fn):
    """Evaluate oracle/em_ref.py in float64 (its casts follow the module-level F / C64)."""
    emo.F, emo.C64 = np.float64, np.complex128
    try:
        with np.errstate(all="ignore"):
            return fn()
    finally:
        emo.F, emo.C64 = np.float32, np.complex64


WEIGHTS = {"a_re": 1e3, "a_im": -7e2, "power": 1e-2, "phase": 1e-3, "length": 0.3, "delay": 2e7, "aoa_az": 1e-2,
           "aoa_el": -2e-2, "aod_az": 3e-2, "aod_el": 1e-2}


def _loss_np(out):
    return (WEIGHTS["a_re"] * out["a"].real + WEIGHTS["a_im"] * out["a"].imag
            + sum(WEIGHTS[k] * out[k] for k in WEIGHTS if k not in ("a_re", "a_im")))


def _loss_torch(out):
    return (WEIGHTS["a_re"] * out["a"].real + WEIGHTS["a_im"] * out["a"].imag
            + sum(WEIGHTS[k] * out[k] for k in WEIGHTS if k not in ("a_re", "a_im"))).sum()


@pytest.mark.parametrize("order", [0, 1, 2, 3])
@pytest.mark.parametrize("polarization", ["V", ("H", (0.3, -0.5, 0.8))])
def test_paths_channel_vjp_vs_central_differences(G, rng, order, polarization):
    """drt_paths_channel_vjp (forward-mode duals in the kernel) vs float64 central differences of the oracle,
    for a weighted sum of all ten outputs, on the valid paths of a box scene with lossy slabs."""
    from differt_amd.plugins import deepmimo

    scene = _box_scene(G, rng)
    mesh = scene.mesh.set_face_materials(np.arange(12) % 3)
    paths = scene.trace_paths(order, compact=True)
    if paths.objects.shape[0] > 40:
        keep = torch.as_tensor(np.sort(rng.choice(paths.objects.shape[0], 40, replace=False)), device="cuda")
        paths = G.TracedPaths(paths.vertices[keep], paths.objects[keep], paths.mask[keep], paths.interaction_types[keep])
    assert paths.objects.shape[0] > 0
    f = 1.2e9
    n_c = emo.complex_refractive_index([5.24, 6.27, 2.0], [0.09, 0.012, 0.0], f)
    n_tab = np.stack((n_c.real, n_c.imag), -1)
    th = np.array([-1.0, 0.05, 0.3], np.float32)
    v = paths.vertices.detach().clone().requires_grad_(True)
    got = deepmimo.paths_channel(G.TracedPaths(v, paths.objects, paths.mask, paths.interaction_types), mesh, n_tab, th,
                                 f, polarization)
    _loss_torch(got).backward()
    grad = _np(v.grad).astype(np.float64)
    assert np.isfinite(grad).all() and np.abs(grad).max() > 0
    V64, obj = _np(paths.vertices).astype(np.float64), _np(paths.objects)
    normals = orc.mesh_normals(orc.triangle_vertices(_np(mesh.vertices), _np(mesh.triangles))).astype(np.float64)
    n64 = _oracle64(lambda: emo.complex_refractive_index(np.array([5.24, 6.27, 2.0]), np.array([0.09, 0.012, 0.0]), f))

    def loss64(Vp):
        return _loss_np(_oracle64(lambda: emo.channel(Vp, obj, normals, np.arange(12) % 3, n64, th.astype(np.float64),
                                                      f, polarization)))

    h = 1e-6
    fd = np.zeros_like(V64)
    for j in range(order + 2):
        for c in range(3):
            up, dn = V64.copy(), V64.copy()
            up[:, j, c] += h
            dn[:, j, c] -= h
            fd[:, j, c] = (loss64(up) - loss64(dn)) / (2 * h)  # paths are independent: one pair of evaluations
    scale = np.abs(fd).max(axis=(1, 2), keepdims=True) + 1e-30
    err = np.abs(grad - fd) / scale
    assert err.max() < 2e-3, float(err.max())       # float32 duals vs float64 differences
    assert np.median(err.max(axis=(1, 2))) < 2e-4


def test_received_power_gradient_end_to_end(G, rng):
    """d(sum of received powers in dBW)/d(tx): tracer VJP + channel VJP chained by autograd, vs float64 central
    differences of (torch restatement of the path vertices) -> (oracle channel)."""
    from differt_amd.plugins import deepmimo
    from oracle import torch_ref

    scene = _box_scene(G, rng, ntx=1, nrx=2)
    mesh = scene.mesh.set_materials("itu_concrete")
    f = 2.4e9
    n_tab, th = deepmimo.material_tables(mesh.material_names, deepmimo.materials, f)
    tx0 = _np(scene.transmitters).astype(np.float64)
    txg = torch.tensor(tx0, dtype=torch.float32, device="cuda", requires_grad=True)
    sc = G.Scene(txg, scene.receivers, mesh)
    paths = sc.trace_paths(2, compact=True)
    assert paths.objects.shape[0] > 3
    out = deepmimo.paths_channel(paths, mesh, n_tab, th, f)
    out["power"].sum().backward()
    g = _np(txg.grad).astype(np.float64)
    obj = _np(paths.objects)
    Vm = torch.tensor(_np(mesh.vertices), dtype=torch.float64)
    Trm = torch.tensor(_np(mesh.triangles), dtype=torch.long)
    rx64 = torch.tensor(_np(scene.receivers), dtype=torch.float64)
    normals = orc.mesh_normals(orc.triangle_vertices(_np(mesh.vertices), _np(mesh.triangles))).astype(np.float64)
    n64 = (n_tab[:, 0] + 1j * n_tab[:, 1]).astype(np.complex128)

    def total_power(txp):
        full = torch_ref.trace_vertices(Vm, Trm, torch.tensor(txp), rx64, torch.tensor(obj[:, 1:-1].astype(np.int64)))
        vv = full[obj[:, 0], obj[:, -1], np.arange(len(obj))].numpy()  # the traced (tx, rx, candidate) triples
        return _oracle64(lambda: emo.channel(vv, obj, normals, np.zeros(12, int), n64, th.astype(np.float64), f))["power"].sum()

    h = 1e-6
    fd = np.zeros_like(tx0)
    for c in range(3):
        up, dn = tx0.copy(), tx0.copy()
        up[0, c] += h
        dn[0, c] -= h
        fd[0, c] = (total_power(up) - total_power(dn)) / (2 * h)
    np.testing.assert_allclose(g, fd, rtol=5e-3, atol=5e-3 * np.abs(fd).max())


def test_example_received_power_ascent():
    """examples/received_power_gradient.py: the gradient steps do increase the received power."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location(
        "received_power_gradient", Path(__file__).resolve().parents[1] / "examples" / "received_power_gradient.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.main(steps=4)
    assert len(hist) == 4 and all(np.isfinite(hist)) and hist[-1] > hist[0]
