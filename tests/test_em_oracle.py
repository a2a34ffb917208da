"""The EM post-processing oracle (oracle/em_ref.py) against the reference's known answers:
differt/tests/em/test_utils.py:33-137 (delays, s/p bases, basis rotations, free-space loss) and
differt/tests/em/test_fresnel.py:32-92 (Fresnel identities, normal incidence, Brewster, total reflection).
"""

from __future__ import annotations

import numpy as np
import pytest

import oracle as orc
from oracle import em_ref as em


def test_path_delay(rng):
    """test_utils.py:33-59."""
    for shape in ((10, 3), (20, 10, 3), (1, 3), (0, 3)):
        path = rng.normal(size=shape).astype(np.float32)
        exp = np.linalg.norm(np.diff(path, axis=-2), axis=-1).sum(-1) / em.c
        np.testing.assert_allclose(em.path_delay(path), exp, rtol=1e-6)
    np.testing.assert_allclose(em.length_to_delay([1.0, 2.0, 4.0], 2.0), [0.5, 1.0, 2.0])


def test_sp_directions():
    """test_utils.py:62-88."""
    cos, sin = np.cos(np.pi / 6), np.sin(np.pi / 6)
    k_i = np.array([[cos, -sin, 0.0], [0.0, -1.0, 0.0]])
    k_r = np.array([[cos, +sin, 0.0], [0.0, +1.0, 0.0]])
    n = np.array([[0.0, 1.0, 0.0], [0.0, 1.0, 0.0]])
    got = em.sp_directions(k_i, k_r, n)
    np.testing.assert_array_equal(got[0][0], got[1][0])
    for (s, p), k in zip(got, (k_i, k_r)):
        np.testing.assert_allclose(np.cross(p, s), k, atol=1e-6)
        np.testing.assert_allclose(np.cross(k, p), s, atol=1e-6)
        np.testing.assert_allclose(np.cross(s, k), p, atol=1e-6)
    np.testing.assert_allclose(got[0][0], [[0, 0, 1], [1, 0, 0]], atol=1e-6)
    np.testing.assert_allclose(got[0][1], [[sin, cos, 0], [0, 0, -1]], atol=1e-6)
    np.testing.assert_allclose(got[1][1], [[-sin, cos, 0], [0, 0, 1]], atol=1e-6)


def _rot_z(a):
    return np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])


def test_sp_rotation_matrix():
    """test_utils.py:91-123."""
    e_i_s, e_i_p = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    got = em.sp_rotation_matrix(e_i_s, e_i_p, [0, 1.0, 0], [-1.0, 0, 0])
    np.testing.assert_allclose(got, _rot_z(-np.pi / 2), atol=1e-7)
    np.testing.assert_allclose(got @ got.T, np.eye(2), atol=1e-7)
    r = np.sqrt(2) / 2
    got = em.sp_rotation_matrix(e_i_s, e_i_p, np.array([1.0, 1, 0]) * r, np.array([-1.0, 1, 0]) * r)
    np.testing.assert_allclose(got, _rot_z(-np.pi / 4), atol=1e-6)
    got = em.sp_rotation_matrix(e_i_s, e_i_p, e_i_s, -e_i_p)  # normal incidence: improper rotation
    np.testing.assert_allclose(np.linalg.det(got), -1.0)
    np.testing.assert_allclose(got, [[1, 0], [0, -1]])


def test_fspl(rng):
    """test_utils.py:126-137."""
    d = rng.uniform(1, 100, (30, 1))
    f = rng.uniform(0.1e9, 10e9, (1, 50))
    got, got_db = em.fspl(d, f), em.fspl(d, f, dB=True)
    np.testing.assert_allclose(10 * np.log10(got), got_db, rtol=1e-5)
    np.testing.assert_allclose(got_db, 20 * np.log10(d) + 20 * np.log10(f) - 147.55, rtol=2e-4)


def test_refractive_index():
    """test_fresnel.py:16-29 (Glass at 1 GHz: eta_r = 6.27, ITU-R P.2040)."""
    np.testing.assert_allclose(em.refractive_index(1.0), 1.0)
    np.testing.assert_allclose(em.refractive_index(6.27), 2.503997, rtol=1e-6)


def test_fresnel_identities(rng):
    """test_fresnel.py:32-56."""
    n_r = (rng.uniform(0.01, 2.0, 100) / rng.uniform(0.01, 2.0, 100)).astype(np.complex64)[:, None]
    theta = np.linspace(0, np.pi / 2, 50)
    ct = np.cos(theta)[None, :]
    (r_s, r_p), (t_s, t_p) = em.fresnel_coefficients(n_r, ct)
    theta_c = np.arcsin(np.minimum(n_r.real, 1.0))
    for arr in (r_s, r_p, t_s, t_p):
        assert np.isfinite(np.where(theta <= theta_c, arr, 0)).all()
    a, b = em.reflection_coefficients(n_r, ct)
    np.testing.assert_array_equal(a, r_s)
    np.testing.assert_array_equal(b, r_p)
    np.testing.assert_allclose(t_s, r_s + 1, atol=1e-6)
    np.testing.assert_allclose(n_r * t_p, r_p + 1, atol=2e-6)


def test_reflection_coefficients_known_answers():
    """test_fresnel.py:59-92."""
    r_s, r_p = em.reflection_coefficients(1.5, 1.0)
    assert r_s == -r_p
    r_s, r_p = em.reflection_coefficients(1.5, np.cos(np.pi / 2))
    np.testing.assert_allclose(r_s**2, -r_p, rtol=1e-6)
    _, r_p = em.reflection_coefficients(np.float32(1.5), np.cos(np.arctan(np.float32(1.5))))
    assert abs(r_p) < 1e-7  # Brewster's angle
    n_r = np.float32(1 / 1.5)
    r_s, r_p = em.reflection_coefficients(n_r, np.cos(np.arcsin(n_r)))
    # total reflection at the critical angle: exact in the reference only because XLA's cos(arcsin(x))
    # makes n^2 + cos^2 - 1 vanish; one ulp off gives sqrt(1e-7) ~ 3e-4 in the coefficients
    np.testing.assert_allclose([r_s, r_p], [1 + 0j, 1 + 0j], atol=5e-3)
    np.testing.assert_allclose(np.abs([r_s, r_p]), 1.0, atol=1e-6)


def test_slab_limits():
    """plugins/deepmimo.py:390-404: negative thickness = half space; zero thickness reflects nothing."""
    n_r, ct = np.complex64(2.2 - 0.1j), np.float32(0.6)
    inf = em.reflection_coefficients(n_r, ct)
    np.testing.assert_array_equal(em.slab_reflection_coefficients(n_r, ct, -1.0, 0.125), inf)
    z = em.slab_reflection_coefficients(n_r, ct, 0.0, 0.125)
    assert abs(z[0]) == 0 and abs(z[1]) == 0
    thick = em.slab_reflection_coefficients(n_r, ct, 50.0, 0.125)  # lossy and thick: the back face is invisible
    np.testing.assert_allclose(thick, inf, rtol=1e-4)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_channel_line_of_sight_and_consistency(order):
    """Identities of the exported quantities (plugins/deepmimo.py:645-711): LOS amplitude is the Friis
    free-space loss, delay = length / c, |a| decreases with every lossy reflection."""
    V, Tr = orc.box_mesh(4.0, 3.0, 2.5, with_top=True)
    nr = orc.mesh_normals(orc.triangle_vertices(V, Tr))
    tx, rx = [[0.7, -0.4, 0.3]], [[-0.9, 0.5, -0.2]]
    cand = orc.generate_all_path_candidates(12, order).astype(np.int32)
    o = orc.trace_path_candidates(V, Tr, tx, rx, cand)
    f = 2.4e9
    n_c = em.complex_refractive_index([5.24], [0.0462 * 2.4**0.7822], f)
    out = em.channel(o["vertices"], o["objects"], nr, np.zeros(12, int), n_c, [-1.0], f)
    m = o["mask"]
    assert m.any()
    np.testing.assert_allclose(out["delay"], out["length"] / em.c, rtol=1e-6)
    np.testing.assert_allclose(out["length"], em.path_length(o["vertices"]), rtol=1e-6)
    friis = (em.c / f) / (4 * np.pi * out["length"])
    mag = np.abs(out["a"])
    if order == 0:
        # V/V antennas: |a| = friis * |<theta_hat(k), theta_hat(-k)>| = friis * sin^2-free projection <= friis
        assert (mag[m] <= friis[m] * (1 + 1e-5)).all() and (mag[m] > 0.05 * friis[m]).all()
    else:
        assert (mag[m] < friis[m]).all()
    np.testing.assert_allclose(out["power"][m], 10 * np.log10(mag[m] ** 2 / em.z_0), rtol=1e-5)
    np.testing.assert_allclose(out["phase"][m], np.degrees(np.angle(out["a"][m])), atol=1e-3)
    # departure / arrival angles point along the first / against the last segment
    d0 = o["vertices"][..., 1, :] - o["vertices"][..., 0, :]
    az = np.degrees(np.arctan2(d0[..., 1], d0[..., 0]))
    np.testing.assert_allclose(out["aod_az"][m], az[m], atol=1e-3)
