// em.hip -- EM post-processing of traced paths (second half), gfx950.
//
//   drt_path_length            reference geometry/_utils.py:150-181 (and em/_utils.py:47-81 path_delay)
//   drt_sp_directions          reference em/_utils.py:84-265
//   drt_sp_rotation_matrix     reference em/_utils.py:268-303
//   drt_fresnel_coefficients   reference em/_fresnel.py:47-214
//   drt_complex_refractive_index (host)  reference plugins/deepmimo.py:480-482
//   drt_paths_channel          reference plugins/deepmimo.py:533-711: per path the complex channel
//                              coefficient (s/p bases per interaction, Fresnel / slab reflection
//                              coefficients, basis rotations, spreading and phase) and the exported
//                              power / phase / delay / angles of departure and arrival.
//
// One lane per path, everything in registers: a path is (order+2) vertices = a few dozen bytes in and
// 48 bytes out, so the kernel is latency-trivial next to the tracer that produced the paths; it exists
// so that the radio quantities come out of the same device-resident pipeline without a host round trip.
// Floating-point row: complex64 arithmetic written out in float32 (tolerance 1e-5, tests/test_em_gpu.py).
#include <complex>

#include "common.hpp"
#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

struct Cf {
    float re, im;
};
__device__ __forceinline__ Cf cf(float re, float im = 0.0f) { return Cf{re, im}; }
__device__ __forceinline__ Cf operator+(Cf a, Cf b) { return Cf{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cf operator-(Cf a, Cf b) { return Cf{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cf operator*(Cf a, Cf b) {
    return Cf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ Cf operator*(Cf a, float s) { return Cf{a.re * s, a.im * s}; }
__device__ __forceinline__ bool is_zero(Cf a) { return a.re == 0.0f && a.im == 0.0f; }
// utils.py:60-67 safe_divide: 0 where the denominator is 0 (Smith's algorithm otherwise)
__device__ __forceinline__ Cf safe_div(Cf a, Cf b) {
    if (is_zero(b)) return Cf{0.0f, 0.0f};
    if (__builtin_fabsf(b.re) >= __builtin_fabsf(b.im)) {
        const float r = b.im / b.re, den = b.re + b.im * r;
        return Cf{(a.re + a.im * r) / den, (a.im - a.re * r) / den};
    }
    const float r = b.re / b.im, den = b.re * r + b.im;
    return Cf{(a.re * r + a.im) / den, (a.im * r - a.re) / den};
}
// principal square root, branch cut on the negative real axis (sign of the imaginary part kept)
__device__ __forceinline__ Cf csqrt_f(Cf z) {
    if (z.re == 0.0f && z.im == 0.0f) return Cf{0.0f, z.im};
    const float m = hypotf(z.re, z.im);
    if (z.re >= 0.0f) {
        const float t = __builtin_sqrtf(0.5f * (m + z.re));
        return Cf{t, z.im / (2.0f * t)};
    }
    const float t = __builtin_sqrtf(0.5f * (m - z.re));
    return Cf{__builtin_fabsf(z.im) / (2.0f * t), __builtin_copysignf(t, z.im)};
}
__device__ __forceinline__ Cf cexp_f(Cf z) {
    const float e = expf(z.re);
    return Cf{e * cosf(z.im), e * sinf(z.im)};
}

__device__ __forceinline__ V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }

// geometry/_utils.py:66-72: v / where(|v| == 0, 1, |v|)
__device__ __forceinline__ V3 normalize_v(V3 v, float &len) {
    len = __builtin_sqrtf(dot(v, v));
    const float s = (len == 0.0f) ? 1.0f : len;
    return V3{v.x / s, v.y / s, v.z / s};
}

// geometry/_utils.py:99-109
__device__ __forceinline__ V3 perpendicular_vector(V3 u) {
    const V3 v = (__builtin_fabsf(u.x) > __builtin_fabsf(u.y)) ? V3{-u.y, u.x, 0.0f} : V3{0.0f, -u.z, u.y};
    float l;
    return normalize_v(cross(u, v), l);
}

struct SpDirs {
    V3 e_i_s, e_i_p, e_r_s, e_r_p;
};
// em/_utils.py:250-265
__device__ __forceinline__ SpDirs sp_directions(V3 k_i, V3 k_r, V3 n) {
    SpDirs r;
    float len, l2;
    r.e_i_s = normalize_v(cross(k_i, n), len);
    if (len == 0.0f) r.e_i_s = perpendicular_vector(k_i);  // normal incidence
    r.e_i_p = normalize_v(cross(r.e_i_s, k_i), l2);
    r.e_r_s = r.e_i_s;
    r.e_r_p = normalize_v(cross(r.e_r_s, k_r), l2);
    return r;
}

struct Fresnel {
    Cf r_s, r_p, t_s, t_p;
};
// em/_fresnel.py:171-214
__device__ __forceinline__ Fresnel fresnel(Cf n_r, float cos_theta_i) {
    const float ct = __builtin_fabsf(cos_theta_i);
    const Cf n2 = n_r * n_r;
    const float ct2 = ct * ct;
    const Cf n2ct = n2 * ct;
    const Cf nct = csqrt_f(Cf{(n2.re + ct2) - 1.0f, n2.im});
    const float two = 2.0f * ct;
    Fresnel f;
    f.r_s = safe_div(Cf{ct - nct.re, -nct.im}, Cf{ct + nct.re, nct.im});
    f.t_s = safe_div(Cf{two, 0.0f}, Cf{ct + nct.re, nct.im});
    f.r_p = safe_div(n2ct - nct, n2ct + nct);
    f.t_p = safe_div(n_r * two, n2ct + nct);
    return f;
}

// plugins/deepmimo.py:390-404: half space for thickness < 0, slab with multiple reflections otherwise
__device__ __forceinline__ void reflection(Cf n_r, float cos_i, float thickness, float wavelength, Cf &r_s,
                                           Cf &r_p) {
    const Fresnel f = fresnel(n_r, cos_i);
    r_s = f.r_s;
    r_p = f.r_p;
    if (thickness >= 0.0f) {
        const Cf eta = n_r * n_r;
        const float sin2 = 1.0f - cos_i * cos_i;
        const Cf a = csqrt_f(Cf{eta.re - sin2, eta.im});
        const float w = (6.2831853071795864769f * thickness) / wavelength;
        const Cf q = a * w;
        const Cf e = cexp_f(Cf{0.0f, -2.0f} * q);
        const Cf one_m_e = Cf{1.0f - e.re, -e.im};
        const Cf ds = (f.r_s * f.r_s) * e, dp = (f.r_p * f.r_p) * e;
        r_s = safe_div(f.r_s * one_m_e, Cf{1.0f - ds.re, -ds.im});
        r_p = safe_div(f.r_p * one_m_e, Cf{1.0f - dp.re, -dp.im});
    }
}

// plugins/deepmimo.py:349-363
__device__ __forceinline__ void spherical_basis(V3 k, V3 &theta_hat, V3 &phi_hat) {
    const float z = fminf(fmaxf(k.z, -1.0f), 1.0f);
    const float theta = acosf(z), phi = atan2f(k.y, k.x);
    const float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
    theta_hat = V3{ct * cp, ct * sp, -st};
    phi_hat = V3{-sp, cp, 0.0f};
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

struct M2 {
    Cf a, b, c, d;  // [[a, b], [c, d]]
};
__device__ __forceinline__ M2 mul(const M2 &x, const M2 &y) {
    return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}

template <int K>
__global__ __launch_bounds__(256) void paths_channel_kernel(
    EmArgs g, const float *__restrict__ vertices, const int32_t *__restrict__ objects, int64_t N,
    float *__restrict__ a_out, float *__restrict__ power, float *__restrict__ phase, float *__restrict__ length,
    float *__restrict__ delay, float *__restrict__ aoa_az, float *__restrict__ aoa_el, float *__restrict__ aod_az,
    float *__restrict__ aod_el) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float *pv = vertices + i * (K + 2) * 3;
    V3 k[K + 1], th[K + 1], ph[K + 1];
    float s_tot = 0.0f;
    V3 prev = ld3(pv);
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        const V3 next = ld3(pv + 3 * (j + 1));
        float s;
        k[j] = normalize_v(next - prev, s);       // :560
        s_tot = s_tot + s;                         // :667
        spherical_basis(k[j], th[j], ph[j]);       // :566
        prev = next;
    }
    // initial field in the (theta, phi) basis of the first segment (:568-589)
    Cf e0, e1;
    if (g.tx_pol == 0) {
        e0 = cf(1.0f);
        e1 = cf(0.0f);
    } else if (g.tx_pol == 1) {
        e0 = cf(0.0f);
        e1 = cf(1.0f);
    } else {
        e0 = cf(dot(g.tx_vec, th[0]));
        e1 = cf(dot(g.tx_vec, ph[0]));
    }
    if constexpr (K > 0) {
        M2 total;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            int64_t tri = objects[i * (K + 2) + 1 + j];
            if (tri < 0 || tri >= g.T) tri = 0;  // padding rows: masked out by the caller
            int64_t mat = g.face_materials[tri];
            if (mat < 0 || mat >= g.M) mat = 0;
            const V3 n = ld3(g.normals + 3 * tri);
            const Cf n_r = Cf{g.n_complex[2 * mat], g.n_complex[2 * mat + 1]};
            const SpDirs d = sp_directions(k[j], k[j + 1], n);                    // :597
            const float cos_i = dot(n, neg(k[j]));                                 // :600
            Cf r_s, r_p;
            reflection(n_r, cos_i, g.thickness[mat], g.wavelength, r_s, r_p);     // :603
            // in_rot = R(theta_in, phi_in -> e_i_s, e_i_p), out_rot = R(e_r_s, e_r_p -> theta_out, phi_out)
            const float i00 = dot(d.e_i_s, th[j]), i01 = dot(d.e_i_s, ph[j]);
            const float i10 = dot(d.e_i_p, th[j]), i11 = dot(d.e_i_p, ph[j]);
            const float o00 = dot(th[j + 1], d.e_r_s), o01 = dot(th[j + 1], d.e_r_p);
            const float o10 = dot(ph[j + 1], d.e_r_s), o11 = dot(ph[j + 1], d.e_r_p);
            const M2 dj{r_s * i00, r_s * i01, r_p * i10, r_p * i11};               // diag(r_s, r_p) @ in_rot
            const M2 jm{dj.a * o00 + dj.c * o01, dj.b * o00 + dj.d * o01,          // out_rot @ ...
                        dj.a * o10 + dj.c * o11, dj.b * o10 + dj.d * o11};
            total = (j == 0) ? jm : mul(jm, total);                                // :633-637
        }
        const Cf n0 = total.a * e0 + total.b * e1, n1 = total.c * e0 + total.d * e1;  // :639
        e0 = n0;
        e1 = n1;
    }
    // projection on the receiver polarisation (:645-664)
    float u0, u1;
    if (g.rx_pol == 2) {
        u0 = dot(g.rx_vec, th[K]);
        u1 = dot(g.rx_vec, ph[K]);
    } else {
        V3 tn, pn;
        spherical_basis(neg(k[K]), tn, pn);
        const float ac = dot(th[K], tn);
        u0 = (g.rx_pol == 0) ? ac : 0.0f;
        u1 = (g.rx_pol == 0) ? 0.0f : -ac;
    }
    Cf a = e0 * u0 + e1 * u1;
    const float spreading = (s_tot == 0.0f) ? 0.0f : 1.0f / s_tot;               // :668
    const float pv_ = (g.phase_k * s_tot) / g.c;                                   // :669
    const Cf shift{cosf(pv_), sinf(pv_)};
    a = a * (shift * spreading);                                                   // :672
    a = a * g.lambda_over_4pi;                                                     // :693
    const float mag = hypotf(a.re, a.im);
    a_out[2 * i] = a.re;
    a_out[2 * i + 1] = a.im;
    power[i] = 10.0f * log10f((mag * mag) / g.z0);                                 // :694-695
    phase[i] = atan2f(a.im, a.re) * kRad2Deg;                                      // :696
    length[i] = s_tot;
    delay[i] = s_tot / g.c;                                                        // :698
    // cartesian_to_spherical of the departure / arrival directions (:699-711)
    const V3 kd = k[0], ka = neg(k[K]);
    float rd = __builtin_sqrtf(dot(kd, kd)), ra = __builtin_sqrtf(dot(ka, ka));
    rd = (rd == 0.0f) ? 1.0f : rd;
    ra = (ra == 0.0f) ? 1.0f : ra;
    aod_el[i] = acosf(kd.z / rd) * kRad2Deg;
    aod_az[i] = atan2f(kd.y, kd.x) * kRad2Deg;
    aoa_el[i] = acosf(ka.z / ra) * kRad2Deg;
    aoa_az[i] = atan2f(ka.y, ka.x) * kRad2Deg;
}

__global__ __launch_bounds__(256) void path_length_kernel(const float *__restrict__ paths, int64_t B, int L,
                                                          float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float *p = paths + i * L * 3;
    float tot = 0.0f;
    for (int j = 0; j + 1 < L; ++j) {
        const V3 v = ld3(p + 3 * (j + 1)) - ld3(p + 3 * j);
        tot = tot + __builtin_sqrtf(dot(v, v));
    }
    out[i] = tot;
}

__global__ __launch_bounds__(256) void sp_directions_kernel(const float *__restrict__ k_i,
                                                            const float *__restrict__ k_r,
                                                            const float *__restrict__ n, int64_t B,
                                                            float *__restrict__ e_i_s, float *__restrict__ e_i_p,
                                                            float *__restrict__ e_r_s, float *__restrict__ e_r_p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const SpDirs d = sp_directions(ld3(k_i + 3 * i), ld3(k_r + 3 * i), ld3(n + 3 * i));
    st3(e_i_s + 3 * i, d.e_i_s);
    st3(e_i_p + 3 * i, d.e_i_p);
    st3(e_r_s + 3 * i, d.e_r_s);
    st3(e_r_p + 3 * i, d.e_r_p);
}

__global__ __launch_bounds__(256) void sp_rotation_kernel(const float *__restrict__ a_s,
                                                          const float *__restrict__ a_p,
                                                          const float *__restrict__ b_s,
                                                          const float *__restrict__ b_p, int64_t B,
                                                          float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const V3 as = ld3(a_s + 3 * i), ap = ld3(a_p + 3 * i), bs = ld3(b_s + 3 * i), bp = ld3(b_p + 3 * i);
    out[4 * i + 0] = dot(bs, as);
    out[4 * i + 1] = dot(bs, ap);
    out[4 * i + 2] = dot(bp, as);
    out[4 * i + 3] = dot(bp, ap);
}

__global__ __launch_bounds__(256) void fresnel_kernel(const float *__restrict__ n_r,
                                                      const float *__restrict__ cos_theta_i, int64_t B,
                                                      float *__restrict__ r_s, float *__restrict__ r_p,
                                                      float *__restrict__ t_s, float *__restrict__ t_p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const Fresnel f = fresnel(Cf{n_r[2 * i], n_r[2 * i + 1]}, cos_theta_i[i]);
    r_s[2 * i] = f.r_s.re, r_s[2 * i + 1] = f.r_s.im;
    r_p[2 * i] = f.r_p.re, r_p[2 * i + 1] = f.r_p.im;
    t_s[2 * i] = f.t_s.re, t_s[2 * i + 1] = f.t_s.im;
    t_p[2 * i] = f.t_p.re, t_p[2 * i + 1] = f.t_p.im;
}

// em/_utils.py:14-44 (length / speed), :345-367 (free-space path loss), em/_fresnel.py:10-44 (sqrt)
__global__ __launch_bounds__(256) void length_to_delay_kernel(const float *__restrict__ len,
                                                              const float *__restrict__ speed, int64_t B,
                                                              float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < B) out[i] = len[i] / speed[i];
}

__global__ __launch_bounds__(256) void fspl_kernel(const float *__restrict__ d, const float *__restrict__ f,
                                                   int64_t B, int db, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    if (db) {
        out[i] = (20.0f * log10f(d[i]) + 20.0f * log10f(f[i])) - 147.55221677811662f;
    } else {
        const float x = ((12.566370614359172f * d[i]) * f[i]) / 299792458.0f;
        out[i] = x * x;
    }
}

__global__ __launch_bounds__(256) void csqrt_kernel(const float *__restrict__ z, int64_t B,
                                                    float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const Cf r = csqrt_f(Cf{z[2 * i], z[2 * i + 1]});
    out[2 * i] = r.re;
    out[2 * i + 1] = r.im;
}

}  // namespace drt

using namespace drt;

extern "C" {

int32_t drt_path_length(const float *paths, int64_t batch, int32_t path_length, float *out, void *stream) {
    DRT_REQUIRE(batch >= 0 && path_length >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(out && (paths || path_length == 0), "null pointer");
    hipLaunchKernelGGL(path_length_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream),
                       paths, batch, (int)path_length, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_sp_directions(const float *k_i, const float *k_r, const float *normals, int64_t batch, float *e_i_s,
                          float *e_i_p, float *e_r_s, float *e_r_p, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(k_i && k_r && normals && e_i_s && e_i_p && e_r_s && e_r_p, "null pointer");
    hipLaunchKernelGGL(sp_directions_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream),
                       k_i, k_r, normals, batch, e_i_s, e_i_p, e_r_s, e_r_p);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_sp_rotation_matrix(const float *e_a_s, const float *e_a_p, const float *e_b_s, const float *e_b_p,
                               int64_t batch, float *out, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(e_a_s && e_a_p && e_b_s && e_b_p && out, "null pointer");
    hipLaunchKernelGGL(sp_rotation_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream),
                       e_a_s, e_a_p, e_b_s, e_b_p, batch, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_fresnel_coefficients(const float *n_r, const float *cos_theta_i, int64_t batch, float *r_s, float *r_p,
                                 float *t_s, float *t_p, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(n_r && cos_theta_i && r_s && r_p && t_s && t_p, "null pointer");
    hipLaunchKernelGGL(fresnel_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream), n_r,
                       cos_theta_i, batch, r_s, r_p, t_s, t_p);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_length_to_delay(const float *length, const float *speed, int64_t batch, float *out, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(length && speed && out, "null pointer");
    hipLaunchKernelGGL(length_to_delay_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream),
                       length, speed, batch, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_fspl(const float *d, const float *f, int64_t batch, int32_t db, float *out, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(d && f && out, "null pointer");
    hipLaunchKernelGGL(fspl_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream), d, f, batch,
                       (int)db, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_refractive_index(const float *epsilon_r, int64_t batch, float *out, void *stream) {
    DRT_REQUIRE(batch >= 0, "negative size");
    if (batch == 0) return DRT_OK;
    DRT_REQUIRE(epsilon_r && out, "null pointer");
    hipLaunchKernelGGL(csqrt_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, as_stream(stream), epsilon_r,
                       batch, out);
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

int32_t drt_complex_refractive_index(const float *eta_r, const float *conductivity, int64_t num_materials,
                                     double frequency, float *n_complex_out) {
    DRT_REQUIRE(num_materials >= 0, "negative size");
    if (num_materials == 0) return DRT_OK;
    DRT_REQUIRE(eta_r && conductivity && n_complex_out, "null pointer");
    DRT_REQUIRE(frequency > 0.0, "frequency must be positive");
    const double omega = 2.0 * 3.14159265358979323846 * frequency;
    const float oe = (float)(omega * 8.8541878128e-12);  // em/_constants.py:7
    for (int64_t i = 0; i < num_materials; ++i) {
        const std::complex<float> n = std::sqrt(std::complex<float>(eta_r[i], -(conductivity[i] / oe)));
        n_complex_out[2 * i] = n.real();
        n_complex_out[2 * i + 1] = n.imag();
    }
    return DRT_OK;
}

int32_t drt_paths_channel(const float *vertices, const int32_t *objects, int64_t num_paths, int32_t order,
                          const float *normals, const int32_t *face_materials, int64_t num_triangles,
                          const float *n_complex, const float *thickness, int64_t num_materials,
                          const drt_em_params *params, float *a, float *power, float *phase, float *length,
                          float *delay, float *aoa_az, float *aoa_el, float *aod_az, float *aod_el, void *stream) {
    DRT_REQUIRE(params, "params is null");
    DRT_REQUIRE(num_paths >= 0 && num_triangles >= 0 && num_materials >= 0, "negative size");
    DRT_REQUIRE(order >= 0 && order <= DRT_MAX_ORDER, "order %d out of range [0, %d]", (int)order, DRT_MAX_ORDER);
    DRT_REQUIRE(params->frequency > 0.0, "frequency must be positive");
    DRT_REQUIRE(params->tx_polarization >= 0 && params->tx_polarization <= 2 && params->rx_polarization >= 0 &&
                    params->rx_polarization <= 2, "polarization must be 0 (V), 1 (H) or 2 (vector)");
    if (num_paths == 0) return DRT_OK;
    DRT_REQUIRE(vertices && objects && a && power && phase && length && delay && aoa_az && aoa_el && aod_az && aod_el,
                "null pointer");
    if (order > 0)
        DRT_REQUIRE(normals && face_materials && n_complex && thickness && num_triangles > 0 && num_materials > 0,
                    "materials / normals are required for order > 0");
    const double c0 = 299792458.0;  // em/_constants.py:1
    EmArgs g{};
    g.normals = normals;
    g.face_materials = face_materials;
    g.T = num_triangles;
    g.n_complex = n_complex;
    g.thickness = thickness;
    g.M = num_materials;
    g.wavelength = (float)(c0 / params->frequency);
    g.lambda_over_4pi = (float)((c0 / params->frequency) / (4.0 * 3.14159265358979323846));
    g.phase_k = (float)(-2.0 * 3.14159265358979323846 * params->frequency);
    g.c = (float)c0;
    g.z0 = (float)376.73031341259;  // em/_constants.py:10
    g.tx_pol = params->tx_polarization;
    g.rx_pol = params->rx_polarization;
    g.tx_vec = V3{params->tx_vector[0], params->tx_vector[1], params->tx_vector[2]};
    g.rx_vec = V3{params->rx_vector[0], params->rx_vector[1], params->rx_vector[2]};
    const dim3 grid((unsigned)ceil_div(num_paths, 256));
#define CALL(K)                                                                                                  \
    hipLaunchKernelGGL(paths_channel_kernel<K>, grid, dim3(256), 0, as_stream(stream), g, vertices, objects,     \
                       num_paths, a, power, phase, length, delay, aoa_az, aoa_el, aod_az, aod_el)
    switch (order) {
        case 0: CALL(0); break;
        case 1: CALL(1); break;
        case 2: CALL(2); break;
        case 3: CALL(3); break;
        case 4: CALL(4); break;
        case 5: CALL(5); break;
        case 6: CALL(6); break;
        case 7: CALL(7); break;
        default: CALL(8); break;
    }
#undef CALL
    DRT_LAUNCH_CHECK();
    return DRT_OK;
}

}  // extern "C"
