// em_core.hpp -- the EM post-processing arithmetic, written once for two scalar types:
//   S = float : the forward kernels (one IEEE rounding per operation, same operation order as
//               oracle/em_ref.py);
//   S = Dual  : forward-mode automatic differentiation (value + one directional derivative), used by
//               the VJP kernel: a path has only 3 (order + 2) input coordinates, so its Jacobian is
//               obtained by 3 (order + 2) dual evaluations inside one lane and contracted with the
//               cotangents in registers -- no hand-derived reverse of the complex Fresnel / slab chain.
// Reference formulas: see em.hip.
#pragma once

#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {
namespace em {

struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual mk(float v, float d = 0.0f) { return Dual{v, d}; }

// ---- value access / construction -------------------------------------------------------------
__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ float val(Dual x) { return x.v; }
template <class S>
__device__ __forceinline__ S lit(float x);
template <>
__device__ __forceinline__ float lit<float>(float x) { return x; }
template <>
__device__ __forceinline__ Dual lit<Dual>(float x) { return Dual{x, 0.0f}; }

// ---- arithmetic on Dual -----------------------------------------------------------------------
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual{-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const float q = a.v / b.v;
    return Dual{q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator+(Dual a, float b) { return Dual{a.v + b, a.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return Dual{a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return Dual{a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return Dual{a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return Dual{a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return Dual{a.v / b, a.d / b}; }
__device__ __forceinline__ Dual operator/(float a, Dual b) {
    const float q = a / b.v;
    return Dual{q, -(q * b.d) / b.v};
}

// ---- elementary functions ---------------------------------------------------------------------
__device__ __forceinline__ float sqrt_s(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ Dual sqrt_s(Dual x) {
    const float s = __builtin_sqrtf(x.v);
    return Dual{s, s == 0.0f ? 0.0f : x.d / (2.0f * s)};
}
__device__ __forceinline__ float sin_s(float x) { return sinf(x); }
__device__ __forceinline__ Dual sin_s(Dual x) { return Dual{sinf(x.v), cosf(x.v) * x.d}; }
__device__ __forceinline__ float cos_s(float x) { return cosf(x); }
__device__ __forceinline__ Dual cos_s(Dual x) { return Dual{cosf(x.v), -sinf(x.v) * x.d}; }
__device__ __forceinline__ float exp_s(float x) { return expf(x); }
__device__ __forceinline__ Dual exp_s(Dual x) {
    const float e = expf(x.v);
    return Dual{e, e * x.d};
}
__device__ __forceinline__ float acos_s(float x) { return acosf(x); }
__device__ __forceinline__ Dual acos_s(Dual x) {
    const float w = 1.0f - x.v * x.v;
    return Dual{acosf(x.v), w > 0.0f ? -x.d / __builtin_sqrtf(w) : 0.0f};
}
__device__ __forceinline__ float atan2_s(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ Dual atan2_s(Dual y, Dual x) {
    const float r2 = x.v * x.v + y.v * y.v;
    return Dual{atan2f(y.v, x.v), r2 > 0.0f ? (x.v * y.d - y.v * x.d) / r2 : 0.0f};
}
__device__ __forceinline__ float hypot_s(float x, float y) { return hypotf(x, y); }
__device__ __forceinline__ Dual hypot_s(Dual x, Dual y) {
    const float h = hypotf(x.v, y.v);
    return Dual{h, h > 0.0f ? (x.v * x.d + y.v * y.d) / h : 0.0f};
}
__device__ __forceinline__ float log10_s(float x) { return log10f(x); }
__device__ __forceinline__ Dual log10_s(Dual x) { return Dual{log10f(x.v), x.d / (x.v * 2.302585092994046f)}; }
__device__ __forceinline__ float abs_s(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ Dual abs_s(Dual x) { return Dual{__builtin_fabsf(x.v), x.v < 0.0f ? -x.d : x.d}; }
__device__ __forceinline__ float clamp1_s(float x) { return fminf(fmaxf(x, -1.0f), 1.0f); }
__device__ __forceinline__ Dual clamp1_s(Dual x) {
    return (x.v < -1.0f) ? Dual{-1.0f, 0.0f} : ((x.v > 1.0f) ? Dual{1.0f, 0.0f} : x);
}
// copysign(|t|, s) with t >= 0: the sign is piecewise constant
__device__ __forceinline__ float copysign_s(float t, float s) { return __builtin_copysignf(t, s); }
__device__ __forceinline__ Dual copysign_s(Dual t, Dual s) {
    const float r = __builtin_copysignf(t.v, s.v);
    return Dual{r, (r < 0.0f) != (t.v < 0.0f) ? -t.d : t.d};
}

// ---- small vectors and complex numbers ---------------------------------------------------------
template <class S>
struct Vec {
    S x, y, z;
};
template <class S>
__device__ __forceinline__ Vec<S> operator-(Vec<S> a, Vec<S> b) { return Vec<S>{a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class S>
__device__ __forceinline__ Vec<S> vneg(Vec<S> a) { return Vec<S>{-a.x, -a.y, -a.z}; }
template <class S>
__device__ __forceinline__ S vdot(Vec<S> a, Vec<S> b) {  // (x0*y0 + x1*y1) + x2*y2
    const S p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z;
    const S s = p0 + p1;
    return s + p2;
}
template <class S>
__device__ __forceinline__ S vdotf(V3 a, Vec<S> b) {  // constant (float) vector with a variable one
    const S p0 = b.x * a.x, p1 = b.y * a.y, p2 = b.z * a.z;
    const S s = p0 + p1;
    return s + p2;
}
template <class S>
__device__ __forceinline__ Vec<S> vcross(Vec<S> a, Vec<S> b) {
    return Vec<S>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class S>
__device__ __forceinline__ Vec<S> vcrossf(Vec<S> a, V3 b) {  // variable x constant
    return Vec<S>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// geometry/_utils.py:66-72: v / where(|v| == 0, 1, |v|)
template <class S>
__device__ __forceinline__ Vec<S> vnormalize(Vec<S> v, S &len) {
    len = sqrt_s(vdot(v, v));
    const S s = (val(len) == 0.0f) ? lit<S>(1.0f) : len;
    return Vec<S>{v.x / s, v.y / s, v.z / s};
}
// geometry/_utils.py:99-109
template <class S>
__device__ __forceinline__ Vec<S> vperpendicular(Vec<S> u) {
    const S z = lit<S>(0.0f);
    const Vec<S> v = (abs_s(val(u.x)) > abs_s(val(u.y))) ? Vec<S>{-u.y, u.x, z} : Vec<S>{z, -u.z, u.y};
    S l;
    return vnormalize(vcross(u, v), l);
}

template <class S>
struct Cx {
    S re, im;
};
template <class S>
__device__ __forceinline__ Cx<S> operator+(Cx<S> a, Cx<S> b) { return Cx<S>{a.re + b.re, a.im + b.im}; }
template <class S>
__device__ __forceinline__ Cx<S> operator-(Cx<S> a, Cx<S> b) { return Cx<S>{a.re - b.re, a.im - b.im}; }
template <class S>
__device__ __forceinline__ Cx<S> operator*(Cx<S> a, Cx<S> b) {
    return Cx<S>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <class S>
__device__ __forceinline__ Cx<S> cscale(Cx<S> a, S s) { return Cx<S>{a.re * s, a.im * s}; }
// utils.py:60-67 safe_divide: 0 where the denominator is 0 (Smith's algorithm otherwise)
template <class S>
__device__ __forceinline__ Cx<S> csafe_div(Cx<S> a, Cx<S> b) {
    if (val(b.re) == 0.0f && val(b.im) == 0.0f) return Cx<S>{lit<S>(0.0f), lit<S>(0.0f)};
    if (__builtin_fabsf(val(b.re)) >= __builtin_fabsf(val(b.im))) {
        const S r = b.im / b.re, den = b.re + b.im * r;
        return Cx<S>{(a.re + a.im * r) / den, (a.im - a.re * r) / den};
    }
    const S r = b.re / b.im, den = b.re * r + b.im;
    return Cx<S>{(a.re * r + a.im) / den, (a.im * r - a.re) / den};
}
// principal square root, branch cut on the negative real axis (sign of the imaginary part kept)
template <class S>
__device__ __forceinline__ Cx<S> csqrt(Cx<S> z) {
    if (val(z.re) == 0.0f && val(z.im) == 0.0f) return Cx<S>{lit<S>(0.0f), z.im};
    const S m = hypot_s(z.re, z.im);
    if (val(z.re) >= 0.0f) {
        const S t = sqrt_s(0.5f * (m + z.re));
        return Cx<S>{t, z.im / (2.0f * t)};
    }
    const S t = sqrt_s(0.5f * (m - z.re));
    return Cx<S>{abs_s(z.im) / (2.0f * t), copysign_s(t, z.im)};
}
template <class S>
__device__ __forceinline__ Cx<S> cexp(Cx<S> z) {
    const S e = exp_s(z.re);
    return Cx<S>{e * cos_s(z.im), e * sin_s(z.im)};
}

// ---- em/_utils.py:250-265 ----------------------------------------------------------------------
template <class S>
struct SpDirsT {
    Vec<S> e_i_s, e_i_p, e_r_s, e_r_p;
};
template <class S>
__device__ __forceinline__ SpDirsT<S> sp_directions_t(Vec<S> k_i, Vec<S> k_r, V3 n) {
    SpDirsT<S> r;
    S len, l2;
    r.e_i_s = vnormalize(vcrossf(k_i, n), len);
    if (val(len) == 0.0f) r.e_i_s = vperpendicular(k_i);  // normal incidence
    r.e_i_p = vnormalize(vcross(r.e_i_s, k_i), l2);
    r.e_r_s = r.e_i_s;
    r.e_r_p = vnormalize(vcross(r.e_r_s, k_r), l2);
    return r;
}

// ---- em/_fresnel.py:171-214 --------------------------------------------------------------------
template <class S>
struct FresnelT {
    Cx<S> r_s, r_p, t_s, t_p;
};
template <class S>
__device__ __forceinline__ FresnelT<S> fresnel_t(Cx<S> n_r, S cos_theta_i) {
    const S ct = abs_s(cos_theta_i);
    const Cx<S> n2 = n_r * n_r;
    const S ct2 = ct * ct;
    const Cx<S> n2ct = cscale(n2, ct);
    const Cx<S> nct = csqrt(Cx<S>{(n2.re + ct2) - 1.0f, n2.im});
    const S two = 2.0f * ct;
    FresnelT<S> f;
    f.r_s = csafe_div(Cx<S>{ct - nct.re, -nct.im}, Cx<S>{ct + nct.re, nct.im});
    f.t_s = csafe_div(Cx<S>{two, lit<S>(0.0f)}, Cx<S>{ct + nct.re, nct.im});
    f.r_p = csafe_div(n2ct - nct, n2ct + nct);
    f.t_p = csafe_div(cscale(n_r, two), n2ct + nct);
    return f;
}

// plugins/deepmimo.py:390-404: half space for thickness < 0, slab with multiple reflections otherwise
template <class S>
__device__ __forceinline__ void reflection_t(Cx<S> n_r, S cos_i, float thickness, float wavelength, Cx<S> &r_s,
                                             Cx<S> &r_p) {
    const FresnelT<S> f = fresnel_t(n_r, cos_i);
    r_s = f.r_s;
    r_p = f.r_p;
    if (thickness >= 0.0f) {
        const Cx<S> eta = n_r * n_r;
        const S sin2 = 1.0f - cos_i * cos_i;
        const Cx<S> a = csqrt(Cx<S>{eta.re - sin2, eta.im});
        const float w = (6.2831853071795864769f * thickness) / wavelength;
        const Cx<S> q = cscale(a, lit<S>(w));
        const Cx<S> e = cexp(Cx<S>{lit<S>(0.0f), lit<S>(-2.0f)} * q);
        const Cx<S> one_m_e = Cx<S>{1.0f - e.re, -e.im};
        const Cx<S> ds = (f.r_s * f.r_s) * e, dp = (f.r_p * f.r_p) * e;
        r_s = csafe_div(f.r_s * one_m_e, Cx<S>{1.0f - ds.re, -ds.im});
        r_p = csafe_div(f.r_p * one_m_e, Cx<S>{1.0f - dp.re, -dp.im});
    }
}

// plugins/deepmimo.py:349-363
template <class S>
__device__ __forceinline__ void spherical_basis_t(Vec<S> k, Vec<S> &theta_hat, Vec<S> &phi_hat) {
    const S z = clamp1_s(k.z);
    const S theta = acos_s(z), phi = atan2_s(k.y, k.x);
    const S st = sin_s(theta), ct = cos_s(theta), sp = sin_s(phi), cp = cos_s(phi);
    theta_hat = Vec<S>{ct * cp, ct * sp, -st};
    phi_hat = Vec<S>{-sp, cp, lit<S>(0.0f)};
}

constexpr float kRad2Deg = 57.295779513082320877f;

struct EmArgs {
    const float *normals;
    const int32_t *face_materials;
    int64_t T;
    const float *n_complex;
    const float *thickness;
    int64_t M;
    float wavelength, lambda_over_4pi, phase_k /* -2 pi f */, c, z0;
    int32_t tx_pol, rx_pol;
    V3 tx_vec, rx_vec;
};

template <class S>
struct M2T {
    Cx<S> a, b, c, d;  // [[a, b], [c, d]]
};
template <class S>
__device__ __forceinline__ M2T<S> mmul(const M2T<S> &x, const M2T<S> &y) {
    return M2T<S>{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}

// The ten per-path outputs, in the order of the cotangent array of the VJP entry point.
template <class S>
struct ChannelOut {
    S a_re, a_im, power, phase, length, delay, aoa_az, aoa_el, aod_az, aod_el;
};

// plugins/deepmimo.py:533-711 for ONE path: `v` = its K + 2 vertices, `obj` = its K + 2 object ids.
template <int K, class S>
__device__ __forceinline__ ChannelOut<S> channel_core(const EmArgs &g, const Vec<S> (&v)[K + 2],
                                                      const int32_t *__restrict__ obj) {
    Vec<S> k[K + 1], th[K + 1], ph[K + 1];
    S s_tot = lit<S>(0.0f);
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        S s;
        k[j] = vnormalize(v[j + 1] - v[j], s);      // :560
        s_tot = s_tot + s;                           // :667
        spherical_basis_t(k[j], th[j], ph[j]);       // :566
    }
    // initial field in the (theta, phi) basis of the first segment (:568-589)
    Cx<S> e0, e1;
    const S zero = lit<S>(0.0f);
    if (g.tx_pol == 0) {
        e0 = Cx<S>{lit<S>(1.0f), zero};
        e1 = Cx<S>{zero, zero};
    } else if (g.tx_pol == 1) {
        e0 = Cx<S>{zero, zero};
        e1 = Cx<S>{lit<S>(1.0f), zero};
    } else {
        e0 = Cx<S>{vdotf(g.tx_vec, th[0]), zero};
        e1 = Cx<S>{vdotf(g.tx_vec, ph[0]), zero};
    }
    if constexpr (K > 0) {
        M2T<S> total;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            int64_t tri = obj[1 + j];
            if (tri < 0 || tri >= g.T) tri = 0;  // padding rows: masked out by the caller
            int64_t mat = g.face_materials[tri];
            if (mat < 0 || mat >= g.M) mat = 0;
            const V3 n = ld3(g.normals + 3 * tri);
            const Cx<S> n_r = Cx<S>{lit<S>(g.n_complex[2 * mat]), lit<S>(g.n_complex[2 * mat + 1])};
            const SpDirsT<S> d = sp_directions_t(k[j], k[j + 1], n);                // :597
            const S cos_i = vdotf(n, vneg(k[j]));                                    // :600
            Cx<S> r_s, r_p;
            reflection_t(n_r, cos_i, g.thickness[mat], g.wavelength, r_s, r_p);     // :603
            // in_rot = R(theta_in, phi_in -> e_i_s, e_i_p), out_rot = R(e_r_s, e_r_p -> theta_out, phi_out)
            const S i00 = vdot(d.e_i_s, th[j]), i01 = vdot(d.e_i_s, ph[j]);
            const S i10 = vdot(d.e_i_p, th[j]), i11 = vdot(d.e_i_p, ph[j]);
            const S o00 = vdot(th[j + 1], d.e_r_s), o01 = vdot(th[j + 1], d.e_r_p);
            const S o10 = vdot(ph[j + 1], d.e_r_s), o11 = vdot(ph[j + 1], d.e_r_p);
            const M2T<S> dj{cscale(r_s, i00), cscale(r_s, i01), cscale(r_p, i10), cscale(r_p, i11)};
            const M2T<S> jm{cscale(dj.a, o00) + cscale(dj.c, o01), cscale(dj.b, o00) + cscale(dj.d, o01),
                            cscale(dj.a, o10) + cscale(dj.c, o11), cscale(dj.b, o10) + cscale(dj.d, o11)};
            total = (j == 0) ? jm : mmul(jm, total);                                 // :633-637
        }
        const Cx<S> n0 = total.a * e0 + total.b * e1, n1 = total.c * e0 + total.d * e1;  // :639
        e0 = n0;
        e1 = n1;
    }
    // projection on the receiver polarisation (:645-664)
    S u0, u1;
    if (g.rx_pol == 2) {
        u0 = vdotf(g.rx_vec, th[K]);
        u1 = vdotf(g.rx_vec, ph[K]);
    } else {
        Vec<S> tn, pn;
        spherical_basis_t(vneg(k[K]), tn, pn);
        const S ac = vdot(th[K], tn);
        u0 = (g.rx_pol == 0) ? ac : zero;
        u1 = (g.rx_pol == 0) ? zero : -ac;
    }
    Cx<S> a = cscale(e0, u0) + cscale(e1, u1);
    const S spreading = (val(s_tot) == 0.0f) ? zero : 1.0f / s_tot;               // :668
    const S pv = (g.phase_k * s_tot) / g.c;                                         // :669
    const Cx<S> shift{cos_s(pv), sin_s(pv)};
    a = a * cscale(shift, spreading);                                               // :672
    a = cscale(a, lit<S>(g.lambda_over_4pi));                                       // :693
    const S mag = hypot_s(a.re, a.im);
    ChannelOut<S> o;
    o.a_re = a.re;
    o.a_im = a.im;
    o.power = 10.0f * log10_s((mag * mag) / g.z0);                                  // :694-695
    o.phase = atan2_s(a.im, a.re) * kRad2Deg;                                       // :696
    o.length = s_tot;
    o.delay = s_tot / g.c;                                                          // :698
    // cartesian_to_spherical of the departure / arrival directions (:699-711)
    const Vec<S> kd = k[0], ka = vneg(k[K]);
    S rd = sqrt_s(vdot(kd, kd)), ra = sqrt_s(vdot(ka, ka));
    rd = (val(rd) == 0.0f) ? lit<S>(1.0f) : rd;
    ra = (val(ra) == 0.0f) ? lit<S>(1.0f) : ra;
    o.aod_el = acos_s(kd.z / rd) * kRad2Deg;
    o.aod_az = atan2_s(kd.y, kd.x) * kRad2Deg;
    o.aoa_el = acos_s(ka.z / ra) * kRad2Deg;
    o.aoa_az = atan2_s(ka.y, ka.x) * kRad2Deg;
    return o;
}

}  // namespace em
}  // namespace drt
