"""CPU restatement (NumPy, float32 / complex64, one rounding per operation) of the EM post-processing
of traced paths -- the second half of SURVEY.md section 8 row f4.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, function by function:
  differt/src/differt/geometry/_utils.py:30-72 (normalize), :76-109 (perpendicular_vector), :150-181 (path_length),
                                         :931-966 (cartesian_to_spherical)
  differt/src/differt/utils.py:37-67 (safe_divide)
  differt/src/differt/em/_utils.py:14-81 (length_to_delay, path_delay), :84-265 (sp_directions),
                                   :268-303 (sp_rotation_matrix), :345-367 (fspl)
  differt/src/differt/em/_fresnel.py:10-44 (refractive_index), :47-214 (fresnel_coefficients)
  differt/src/differt/plugins/deepmimo.py:334-363 (_spherical_basis), :367-404 (_get_reflection_coefficients),
                                          :480-488 (complex refractive index of the materials),
                                          :533-693 (per-path channel coefficient and the exported quantities)
Pinned by the reference's known answers for these functions (differt/tests/em/test_utils.py:62-137,
test_fresnel.py:32-92) in tests/test_em_oracle.py.  The reference holds NO numeric golden for the
DeepMIMO export itself (its Sionna comparison needs a downloaded scene): that part is "parity unpinned"
beyond the identities checked there.
"""

from __future__ import annotations

import numpy as np

F = np.float32
C64 = np.complex64
c = 299792458.0            # em/_constants.py:1
epsilon_0 = 8.8541878128e-12  # em/_constants.py:7
z_0 = 376.73031341259      # em/_constants.py:10


def _f(x):
    return np.asarray(x, dtype=F)


def _dot(a, b):
    p = (a * b).astype(F)
    return ((p[..., 0] + p[..., 1]).astype(F) + p[..., 2]).astype(F)


def _cross(a, b):
    x = ((a[..., 1] * b[..., 2]).astype(F) - (a[..., 2] * b[..., 1]).astype(F)).astype(F)
    y = ((a[..., 2] * b[..., 0]).astype(F) - (a[..., 0] * b[..., 2]).astype(F)).astype(F)
    z = ((a[..., 0] * b[..., 1]).astype(F) - (a[..., 1] * b[..., 0]).astype(F)).astype(F)
    return np.stack((x, y, z), axis=-1)


def normalize(v, keepdims=False):
    """geometry/_utils.py:66-72."""
    v = _f(v)
    ln = np.sqrt(_dot(v, v))[..., None]
    safe = np.where(ln == 0, F(1), ln)
    return (v / safe).astype(F), (ln if keepdims else ln[..., 0])


def perpendicular_vector(u):
    """geometry/_utils.py:99-109."""
    u = _f(u)
    z = np.zeros_like(u[..., 0])
    v = np.where((np.abs(u[..., 0]) > np.abs(u[..., 1]))[..., None],
                 np.stack((-u[..., 1], u[..., 0], z), axis=-1),
                 np.stack((z, -u[..., 2], u[..., 1]), axis=-1))
    return normalize(_cross(u, v))[0]


def _cdiv(a, b):
    """Complex division in float32 with Smith's algorithm (a real / real division stays ONE correctly
    rounded division, which the reference's exact known answers rely on, test_fresnel.py:59-92)."""
    a, b = np.broadcast_arrays(np.asarray(a).astype(C64), np.asarray(b).astype(C64))
    ar, ai, br, bi = a.real.astype(F), a.imag.astype(F), b.real.astype(F), b.imag.astype(F)
    big = np.abs(br) >= np.abs(bi)
    with np.errstate(divide="ignore", invalid="ignore"):
        r1 = (bi / br).astype(F)
        d1 = (br + (bi * r1).astype(F)).astype(F)
        re1 = ((ar + (ai * r1).astype(F)).astype(F) / d1).astype(F)
        im1 = ((ai - (ar * r1).astype(F)).astype(F) / d1).astype(F)
        r2 = (br / bi).astype(F)
        d2 = ((br * r2).astype(F) + bi).astype(F)
        re2 = (((ar * r2).astype(F) + ai).astype(F) / d2).astype(F)
        im2 = (((ai * r2).astype(F) - ar).astype(F) / d2).astype(F)
    return (np.where(big, re1, re2) + 1j * np.where(big, im1, im2)).astype(C64)


def safe_divide(num, den):
    """utils.py:60-67."""
    num, den = np.asarray(num), np.asarray(den)
    zero = den == 0
    safe = np.where(zero, np.ones_like(den), den)
    if np.iscomplexobj(num) or np.iscomplexobj(den):
        out = _cdiv(num, safe)
    else:
        out = num / safe
    return np.where(zero, np.zeros_like(out), out).astype(np.result_type(num, den))


def path_length(path):
    """geometry/_utils.py:176-181."""
    v = np.diff(_f(path), axis=-2).astype(F)
    ln = np.sqrt(_dot(v, v))
    out = np.zeros(ln.shape[:-1], F)
    for j in range(ln.shape[-1]):
        out = (out + ln[..., j]).astype(F)
    return out


def length_to_delay(length, speed=c):
    return (_f(length) / _f(speed)).astype(F)


def path_delay(path, speed=c):
    return length_to_delay(path_length(path), speed)


def fspl(d, f, dB=False):  # noqa: N803
    """em/_utils.py:364-367."""
    d, f = _f(d), _f(f)
    if dB:
        return (F(20) * np.log10(d) + F(20) * np.log10(f) - F(147.55221677811662)).astype(F)
    x = (F(4 * np.pi) * d * f / F(c)).astype(F)
    return (x * x).astype(F)


def sp_directions(k_i, k_r, normals):
    """em/_utils.py:250-265."""
    k_i, k_r, n = np.broadcast_arrays(_f(k_i), _f(k_r), _f(normals))
    e_i_s, norm = normalize(_cross(k_i, n), keepdims=True)
    e_i_s = np.where(norm == 0, perpendicular_vector(k_i), e_i_s)
    e_i_p = normalize(_cross(e_i_s, k_i))[0]
    e_r_s = e_i_s
    e_r_p = normalize(_cross(e_r_s, k_r))[0]
    return (e_i_s, e_i_p), (e_r_s, e_r_p)


def sp_rotation_matrix(e_a_s, e_a_p, e_b_s, e_b_p):
    """em/_utils.py:292-303."""
    e_a_s, e_a_p, e_b_s, e_b_p = np.broadcast_arrays(_f(e_a_s), _f(e_a_p), _f(e_b_s), _f(e_b_p))
    r = np.stack((_dot(e_b_s, e_a_s), _dot(e_b_s, e_a_p), _dot(e_b_p, e_a_s), _dot(e_b_p, e_a_p)), axis=-1)
    return r.reshape(*r.shape[:-1], 2, 2)


def refractive_index(epsilon_r, mu_r=None):
    return np.sqrt(np.asarray(epsilon_r) if mu_r is None else np.asarray(epsilon_r) * np.asarray(mu_r))


def fresnel_coefficients(n_r, cos_theta_i):
    """em/_fresnel.py:171-214; complex64 throughout."""
    n_r = np.asarray(n_r).astype(C64)
    ct = np.abs(_f(cos_theta_i))
    n2 = (n_r * n_r).astype(C64)
    ct2 = (ct * ct).astype(F)
    n2ct = (n2 * ct).astype(C64)
    nct = np.sqrt(((n2 + ct2).astype(C64) - F(1)).astype(C64)).astype(C64)
    two = (F(2) * ct).astype(F)
    r_s = safe_divide((ct - nct).astype(C64), (ct + nct).astype(C64)).astype(C64)
    t_s = safe_divide(two.astype(C64), (ct + nct).astype(C64)).astype(C64)
    r_p = safe_divide((n2ct - nct).astype(C64), (n2ct + nct).astype(C64)).astype(C64)
    t_p = safe_divide((n_r * two).astype(C64), (n2ct + nct).astype(C64)).astype(C64)
    return (r_s, r_p), (t_s, t_p)


def reflection_coefficients(n_r, cos_theta_i):
    return fresnel_coefficients(n_r, cos_theta_i)[0]


def refraction_coefficients(n_r, cos_theta_i):
    return fresnel_coefficients(n_r, cos_theta_i)[1]


def slab_reflection_coefficients(n_r, cos_theta_i, thickness, wavelength):
    """plugins/deepmimo.py:390-404: infinite half-space when thickness < 0, slab otherwise."""
    n_r = np.asarray(n_r).astype(C64)
    ct = _f(cos_theta_i)
    th = _f(thickness)
    r_s_inf, r_p_inf = reflection_coefficients(n_r, ct)
    eta = (n_r * n_r).astype(C64)
    sin2 = (F(1) - (ct * ct).astype(F)).astype(F)
    a = np.sqrt((eta - sin2).astype(C64)).astype(C64)
    q = (((F(2 * np.pi) * th).astype(F) / F(wavelength)).astype(F) * a).astype(C64)
    e = np.exp((C64(-2j) * q).astype(C64)).astype(C64)
    one = F(1)
    r_s_slab = safe_divide((r_s_inf * (one - e)).astype(C64), (one - (r_s_inf * r_s_inf).astype(C64) * e).astype(C64))
    r_p_slab = safe_divide((r_p_inf * (one - e)).astype(C64), (one - (r_p_inf * r_p_inf).astype(C64) * e).astype(C64))
    use = th >= 0
    return np.where(use, r_s_slab, r_s_inf).astype(C64), np.where(use, r_p_slab, r_p_inf).astype(C64)


def complex_refractive_index(eta_r, conductivity, frequency):
    """plugins/deepmimo.py:480-482."""
    omega = 2.0 * np.pi * frequency
    eps = (_f(eta_r) - C64(1j) * (_f(conductivity) / F(omega * epsilon_0)).astype(F)).astype(C64)
    return np.sqrt(eps).astype(C64)


def spherical_basis(k):
    """plugins/deepmimo.py:349-363."""
    k = _f(k)
    x, y = k[..., 0], k[..., 1]
    z = np.clip(k[..., 2], F(-1), F(1))
    theta = np.arccos(z).astype(F)
    phi = np.arctan2(y, x).astype(F)
    st, ct, sp, cp = (fn(a).astype(F) for fn, a in ((np.sin, theta), (np.cos, theta), (np.sin, phi), (np.cos, phi)))
    theta_hat = np.stack(((ct * cp).astype(F), (ct * sp).astype(F), -st), axis=-1)
    phi_hat = np.stack((-sp, cp, np.zeros_like(phi)), axis=-1)
    return theta_hat, phi_hat


def cartesian_to_spherical(xyz):
    """geometry/_utils.py:958-966."""
    xyz = _f(xyz)
    r = np.sqrt(_dot(xyz, xyz))
    r = np.where(r == 0, F(1), r)
    p = np.arccos((xyz[..., 2] / r).astype(F)).astype(F)
    a = np.arctan2(xyz[..., 1], xyz[..., 0]).astype(F)
    return np.stack((r, p, a), axis=-1)


def _pol(pol, theta_hat, phi_hat):
    if isinstance(pol, str):
        one, zero = np.ones(theta_hat.shape[:-1], F), np.zeros(theta_hat.shape[:-1], F)
        return (one, zero) if pol == "V" else (zero, one)
    p = _f(pol)
    return _dot(np.broadcast_to(p, theta_hat.shape), theta_hat), _dot(np.broadcast_to(p, phi_hat.shape), phi_hat)


def channel(vertices, objects, normals, face_materials, n_complex, thickness, frequency, polarization="V"):
    """Per-path quantities of `deepmimo.export` (plugins/deepmimo.py:533-711) for paths
    `vertices f32[*B, k+2, 3]`, `objects i32[*B, k+2]`: returns a dict with the complex coefficient
    `a` (after the lambda/4pi factor, :693), `power` [dBW], `phase` [deg], `delay` [s], `length`,
    `aoa_az, aoa_el, aod_az, aod_el` [deg].  All reflections (interaction type 0)."""
    v = _f(vertices)
    objects = np.asarray(objects)
    order = v.shape[-2] - 2
    tx_pol, rx_pol = polarization if isinstance(polarization, tuple) and len(polarization) == 2 else (polarization,) * 2
    seg = np.diff(v, axis=-2).astype(F)
    k, s = normalize(seg, keepdims=True)                       # :560
    theta_hat, phi_hat = spherical_basis(k)                     # :566
    e0, e1 = _pol(tx_pol, theta_hat[..., 0, :], phi_hat[..., 0, :])  # :568-589
    e = np.stack((e0, e1), axis=-1).astype(C64)
    wavelength = c / frequency
    if order > 0:
        obj = objects[..., 1:-1]
        mat = np.asarray(face_materials)[obj]
        n = _f(normals)[obj]
        k_in, k_out = k[..., :-1, :], k[..., 1:, :]
        n_r = np.asarray(n_complex).astype(C64)[mat]
        th = _f(thickness)[mat]
        (e_i_s, e_i_p), (e_r_s, e_r_p) = sp_directions(k_in, k_out, n)    # :597
        cos_i = _dot(n, -k_in)                                               # :600
        r_s, r_p = slab_reflection_coefficients(n_r, cos_i, th, wavelength)  # :603
        in_rot = sp_rotation_matrix(theta_hat[..., :-1, :], phi_hat[..., :-1, :], e_i_s, e_i_p)   # :615
        out_rot = sp_rotation_matrix(e_r_s, e_r_p, theta_hat[..., 1:, :], phi_hat[..., 1:, :])    # :616
        def cmul2(x, y):  # 2x2 complex matrix product, entries as (m00, m01, m10, m11)
            return (((x[0] * y[0]).astype(C64) + (x[1] * y[2]).astype(C64)).astype(C64),
                    ((x[0] * y[1]).astype(C64) + (x[1] * y[3]).astype(C64)).astype(C64),
                    ((x[2] * y[0]).astype(C64) + (x[3] * y[2]).astype(C64)).astype(C64),
                    ((x[2] * y[1]).astype(C64) + (x[3] * y[3]).astype(C64)).astype(C64))

        j_total = None
        for j in range(order):  # J_j = out_rot_j @ (diag(r_s, r_p) @ in_rot_j)   (:619-630)
            a_, b_ = in_rot[..., j, :, :].astype(C64), out_rot[..., j, :, :].astype(C64)
            d = ((r_s[..., j] * a_[..., 0, 0]).astype(C64), (r_s[..., j] * a_[..., 0, 1]).astype(C64),
                 (r_p[..., j] * a_[..., 1, 0]).astype(C64), (r_p[..., j] * a_[..., 1, 1]).astype(C64))
            jm = cmul2((b_[..., 0, 0], b_[..., 0, 1], b_[..., 1, 0], b_[..., 1, 1]), d)
            j_total = jm if j_total is None else cmul2(jm, j_total)  # reduce(lambda x, y: y @ x)  (:633-637)
        e = np.stack((((j_total[0] * e[..., 0]).astype(C64) + (j_total[1] * e[..., 1]).astype(C64)).astype(C64),
                      ((j_total[2] * e[..., 0]).astype(C64) + (j_total[3] * e[..., 1]).astype(C64)).astype(C64)),
                     axis=-1)  # :639
    th_last, ph_last = theta_hat[..., -1, :], phi_hat[..., -1, :]
    if isinstance(rx_pol, str):                                                # :649-657
        th_neg = spherical_basis(-k[..., -1, :])[0]
        ac = _dot(th_last, th_neg)
        u = (ac, np.zeros_like(ac)) if rx_pol == "V" else (np.zeros_like(ac), -ac)
    else:
        u = _pol(rx_pol, th_last, ph_last)
    a = ((u[0] * e[..., 0]).astype(C64) + (u[1] * e[..., 1]).astype(C64)).astype(C64)   # :664
    s_tot = np.zeros(s.shape[:-2] + (1,), F)
    for j in range(s.shape[-2]):
        s_tot = (s_tot + s[..., j, :]).astype(F)
    s_tot = s_tot[..., 0]
    spreading = safe_divide(F(1), s_tot).astype(F)                                 # :668
    phase_val = (((F(-2.0 * np.pi * frequency) * s_tot).astype(F)) / F(c)).astype(F)    # :669
    shift = (np.cos(phase_val).astype(F) + 1j * np.sin(phase_val).astype(F)).astype(C64)
    a = (a * (spreading * shift).astype(C64)).astype(C64)                           # :672
    a = (a * F(wavelength / (4 * np.pi))).astype(C64)                                # :693
    mag = np.abs(a).astype(F)
    with np.errstate(divide="ignore"):
        power = (F(10) * np.log10(((mag * mag).astype(F) / F(z_0)).astype(F))).astype(F)   # :694-695
    phase = np.degrees(np.arctan2(a.imag, a.real).astype(F)).astype(F)                # :696
    k_d, k_a = k[..., 0, :], -k[..., -1, :]
    sd, sa = cartesian_to_spherical(k_d), cartesian_to_spherical(k_a)
    return {
        "a": a, "power": power, "phase": phase, "length": s_tot, "delay": (s_tot / F(c)).astype(F),
        "aoa_az": np.degrees(sa[..., 2]).astype(F), "aoa_el": np.degrees(sa[..., 1]).astype(F),
        "aod_az": np.degrees(sd[..., 2]).astype(F), "aod_el": np.degrees(sd[..., 1]).astype(F),
    }
