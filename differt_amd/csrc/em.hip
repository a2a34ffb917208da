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
#include "em_core.hpp"
#include "geom.hpp"

#pragma clang fp contract(off)

namespace drt {

using namespace em;

__device__ __forceinline__ Vec<float> ldv(const float *p) { return Vec<float>{p[0], p[1], p[2]}; }
__device__ __forceinline__ void stv(float *p, Vec<float> a) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
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
    Vec<float> v[K + 2];
#pragma unroll
    for (int j = 0; j < K + 2; ++j) v[j] = ldv(pv + 3 * j);
    const ChannelOut<float> o = channel_core<K, float>(g, v, objects + i * (K + 2));
    a_out[2 * i] = o.a_re;
    a_out[2 * i + 1] = o.a_im;
    power[i] = o.power;
    phase[i] = o.phase;
    length[i] = o.length;
    delay[i] = o.delay;
    aoa_az[i] = o.aoa_az;
    aoa_el[i] = o.aoa_el;
    aod_az[i] = o.aod_az;
    aod_el[i] = o.aod_el;
}

// VJP with respect to the path vertices: forward-mode duals, one evaluation per input coordinate, the
// Jacobian column contracted with the ten cotangents (ChannelOut order) in registers.  Zero cotangents
// are skipped, so padding / masked paths (whose outputs may be non-finite) contribute exactly zero.
template <int K>
__global__ __launch_bounds__(256) void paths_channel_vjp_kernel(EmArgs g, const float *__restrict__ vertices,
                                                                const int32_t *__restrict__ objects, int64_t N,
                                                                const float *__restrict__ cot,
                                                                float *__restrict__ g_vertices) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float c[10];
    bool any = false;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        c[q] = cot[10 * i + q];
        any = any || (c[q] != 0.0f);
    }
    float *gv = g_vertices + i * (K + 2) * 3;
    if (!any) {
        for (int d = 0; d < 3 * (K + 2); ++d) gv[d] = 0.0f;
        return;
    }
    const float *pv = vertices + i * (K + 2) * 3;
    for (int d = 0; d < 3 * (K + 2); ++d) {
        Vec<Dual> v[K + 2];
#pragma unroll
        for (int j = 0; j < K + 2; ++j)
            v[j] = Vec<Dual>{mk(pv[3 * j], d == 3 * j ? 1.0f : 0.0f), mk(pv[3 * j + 1], d == 3 * j + 1 ? 1.0f : 0.0f),
                             mk(pv[3 * j + 2], d == 3 * j + 2 ? 1.0f : 0.0f)};
        const ChannelOut<Dual> o = channel_core<K, Dual>(g, v, objects + i * (K + 2));
        const float dv[10] = {o.a_re.d, o.a_im.d, o.power.d, o.phase.d, o.length.d,
                              o.delay.d, o.aoa_az.d, o.aoa_el.d, o.aod_az.d, o.aod_el.d};
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 10; ++q)
            if (c[q] != 0.0f) acc += c[q] * dv[q];
        gv[d] = acc;
    }
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
    const SpDirsT<float> d = sp_directions_t<float>(ldv(k_i + 3 * i), ldv(k_r + 3 * i), ld3(n + 3 * i));
    stv(e_i_s + 3 * i, d.e_i_s);
    stv(e_i_p + 3 * i, d.e_i_p);
    stv(e_r_s + 3 * i, d.e_r_s);
    stv(e_r_p + 3 * i, d.e_r_p);
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
    const FresnelT<float> f = fresnel_t<float>(Cx<float>{n_r[2 * i], n_r[2 * i + 1]}, cos_theta_i[i]);
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
    const Cx<float> r = csqrt<float>(Cx<float>{z[2 * i], z[2 * i + 1]});
    out[2 * i] = r.re;
    out[2 * i + 1] = r.im;
}

}  // namespace drt

using namespace drt;
using namespace drt::em;

static EmArgs make_em_args(const float *normals, const int32_t *face_materials, int64_t num_triangles,
                           const float *n_complex, const float *thickness, int64_t num_materials,
                           const drt_em_params *params) {
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
    return g;
}

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
    const EmArgs g = make_em_args(normals, face_materials, num_triangles, n_complex, thickness, num_materials, params);
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

int32_t drt_paths_channel_vjp(const float *vertices, const int32_t *objects, int64_t num_paths, int32_t order,
                              const float *normals, const int32_t *face_materials, int64_t num_triangles,
                              const float *n_complex, const float *thickness, int64_t num_materials,
                              const drt_em_params *params, const float *cotangents, float *grad_vertices,
                              void *stream) {
    DRT_REQUIRE(params, "params is null");
    DRT_REQUIRE(num_paths >= 0 && num_triangles >= 0 && num_materials >= 0, "negative size");
    DRT_REQUIRE(order >= 0 && order <= DRT_MAX_ORDER, "order %d out of range [0, %d]", (int)order, DRT_MAX_ORDER);
    DRT_REQUIRE(params->frequency > 0.0, "frequency must be positive");
    if (num_paths == 0) return DRT_OK;
    DRT_REQUIRE(vertices && objects && cotangents && grad_vertices, "null pointer");
    if (order > 0)
        DRT_REQUIRE(normals && face_materials && n_complex && thickness && num_triangles > 0 && num_materials > 0,
                    "materials / normals are required for order > 0");
    const EmArgs g = make_em_args(normals, face_materials, num_triangles, n_complex, thickness, num_materials, params);
    const dim3 grid((unsigned)ceil_div(num_paths, 256));
#define CALL(K)                                                                                                \
    hipLaunchKernelGGL(paths_channel_vjp_kernel<K>, grid, dim3(256), 0, as_stream(stream), g, vertices, objects, \
                       num_paths, cotangents, grad_vertices)
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
